"""gogp_amd -- MI355X-native (gfx950) implementation of the GP-regression hot
path of infergo-ml/gogp behind the reference's own API surface.

  gogp_amd.kernel  mirrors the reference's ``kernel`` package (kernel/kernel.go, kernel/noise.go)
  gogp_amd.gp      mirrors the reference's ``gp`` package     (gp/gp.go, gp/model.go)

Everything numeric runs in libgogp_hip.so (hand-written HIP, C ABI in
include/gogp_hip.h).  There is no CPU fallback; importing ``gogp_amd.gp`` and
constructing a ``GP`` raises if the library or a HIP device is missing.
"""
from . import kernel  # noqa: F401

__all__ = ["kernel", "gp"]
__version__ = "0.1.0"
