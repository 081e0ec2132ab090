"""ctypes loader of libgogp_hip.so (the C ABI of include/gogp_hip.h).

There is deliberately NO fallback: if the shared library is missing or a
symbol cannot be bound, importing the binding raises.  Nothing under oracle/
is ever imported from here.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

from .kernel import CDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgogp_hip.so")
HOOKS_PATH = os.path.join(_HERE, "libgogp_testhooks.so")

GOGP_OK, GOGP_EARG, GOGP_ENOTPD, GOGP_EHIP, GOGP_ESTATE, GOGP_ENOMEM, GOGP_ECOND = 0, 1, 2, 3, 4, 5, 6
GOGP_MAX_CANDIDATES = 16

#: every symbol include/gogp_hip.h declares: (name, restype, argtypes)
_dp = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.c_int64
_h = ctypes.c_void_p
_descp = ctypes.POINTER(CDesc)
#: callback types of the sharded evaluation (include/gogp_hip.h)
GOGP_UNIQUE_ID_BYTES = 128


class CXfer(ctypes.Structure):
    _fields_ = [("peer", ctypes.c_int32), ("is_send", ctypes.c_int32), ("buf", ctypes.c_void_p),
                ("bytes", ctypes.c_int64)]


EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(CXfer), ctypes.c_int32)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                                ctypes.c_int64)
SYMBOLS = [
    ("gogp_desc_check", ctypes.c_int, [_descp]),
    ("gogp_desc_ntheta_noise", ctypes.c_int, [_descp]),
    ("gogp_create", ctypes.c_int, [_descp, ctypes.c_int, ctypes.POINTER(_h)]),
    ("gogp_destroy", None, [_h]),
    ("gogp_last_error", ctypes.c_char_p, [_h]),
    ("gogp_notpd_index", _i64, [_h]),
    ("gogp_graph_info", ctypes.c_int, [_h, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]),
    ("gogp_set_data", ctypes.c_int, [_h, _dp, _dp, _i64]),
    ("gogp_set_data_device", ctypes.c_int, [_h, ctypes.c_void_p, ctypes.c_void_p, _i64]),
    ("gogp_absorb", ctypes.c_int, [_h, _dp, _dp]),
    ("gogp_observe", ctypes.c_int, [_h, _dp, _i64, _dp]),
    ("gogp_observe_full", ctypes.c_int, [_h, _dp, _i64, _dp]),
    ("gogp_lml", ctypes.c_int, [_h, _dp]),
    ("gogp_gradient", ctypes.c_int, [_h, _dp, _i64]),
    ("gogp_observe_gradient_batch", ctypes.c_int,
     [ctypes.POINTER(_h), ctypes.c_int, _dp, _i64, _dp, _dp, ctypes.POINTER(ctypes.c_int)]),
    ("gogp_observe_gradient_candidates", ctypes.c_int,
     [_h, ctypes.c_int, _dp, _i64, _dp, _dp, ctypes.POINTER(ctypes.c_int)]),
    ("gogp_produce", ctypes.c_int, [_h, _dp, _i64, _dp, _dp]),
    ("gogp_n", _i64, [_h]),
    ("gogp_get_alpha", ctypes.c_int, [_h, _dp]),
    ("gogp_get_factor", ctypes.c_int, [_h, _dp]),
    ("gogp_get_factor_rows", ctypes.c_int, [_h, ctypes.POINTER(_i64), _i64, _dp]),
    ("gogp_get_factor_diag", ctypes.c_int, [_h, _dp]),
    ("gogp_set_factor", ctypes.c_int, [_h, _dp, _dp, _dp, _dp]),
    ("gogp_dist_grid", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    ("gogp_dist_unique_id", ctypes.c_int, [ctypes.c_void_p]),
    ("gogp_dist_init_rccl", ctypes.c_int,
     [_h, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    ("gogp_dist_init_callbacks", ctypes.c_int,
     [_h, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
      ctypes.c_void_p]),
    ("gogp_dist_local_bytes", _i64, [_h]),
    ("gogp_dist_comm_ranks", ctypes.c_int, [_h, ctypes.POINTER(ctypes.c_int)]),
    ("gogp_dist_selftest", ctypes.c_int, [_h, ctypes.c_int, _i64]),
    ("gogp_profile_enable", ctypes.c_int, [_h, ctypes.c_int]),
    ("gogp_profile_read", ctypes.c_int, [_h, _dp, ctypes.POINTER(_i64), _dp, _dp]),
    ("gogp_profile_read_launches", ctypes.c_int,
     [_h, _i64, _dp, _dp, _dp, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    ("gogp_profile_read_aux", ctypes.c_int, [_h, ctypes.c_int, _dp, ctypes.POINTER(_i64)]),
    ("gogp_set_option", ctypes.c_int, [_h, ctypes.c_char_p, _i64]),
    ("gogp_version", ctypes.c_char_p, []),
]

#: every symbol include/gogp_testhooks.h declares (libgogp_testhooks.so: measurement and
#: diagnostic hooks for tests/, tools/ and bench.py -- not part of the product ABI)
HOOK_SYMBOLS = [
    ("gogp_mfma_f64_peak", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]),
    ("gogp_mfma_f32_peak", ctypes.c_int, [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]),
    ("gogp_bench_gemm", ctypes.c_int,
     [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _i64, ctypes.c_int, _dp, _dp]),
    ("gogp_test_diag256", ctypes.c_int,
     [ctypes.c_int, _dp, _dp, _dp, ctypes.POINTER(ctypes.c_uint64), _dp]),
    ("gogp_test_dgemm_nt", ctypes.c_int,
     [ctypes.c_int, _i64, _i64, _i64, ctypes.c_double, _dp, _dp, ctypes.c_double, _dp]),
    ("gogp_test_valu_cost", ctypes.c_int, [ctypes.c_int, _dp]),
    ("gogp_test_panel128", ctypes.c_int,
     [ctypes.c_int, _dp, _dp, _i64, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), _dp]),
    ("gogp_test_panel128_slabs", ctypes.c_int, [ctypes.c_int, _dp, _dp, _i64, ctypes.c_int, ctypes.c_int, _dp]),
    # per-rank replay of the sharded sweep (tools/sharded_replay.py): a transport that reads recorded panels
    ("gogp_test_dist_init_replay", ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 4),
]


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into gogp_amd/libgogp_hip.so
    (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j8", "-s"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce %s" % LIB_PATH)
    return LIB_PATH


_lib = None
_hooks = None


def hooks() -> ctypes.CDLL:
    """The measurement-hook library (include/gogp_testhooks.h)."""
    global _hooks
    if _hooks is None:
        lib()  # the hook library links the product library
        if not os.path.exists(HOOKS_PATH):
            raise ImportError("%s is missing: run __graft_entry__.build()" % HOOKS_PATH)
        L = ctypes.CDLL(HOOKS_PATH)
        for name, restype, argtypes in HOOK_SYMBOLS:
            f = getattr(L, name)
            f.restype = restype
            f.argtypes = argtypes
        _hooks = L
    return _hooks


def lib() -> ctypes.CDLL:
    """The bound library.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(gogp_amd has no CPU fallback)" % LIB_PATH)
        # PyTorch's wheel bundles its own copies of the HIP runtime and RCCL whose NEEDED names
        # lack the version suffix: if it is imported AFTER this library, the loader does not
        # match them with the already-loaded /opt/rocm copies and the process ends up with two
        # HIP runtimes (observed: heap corruption at exit).  Where torch exists, load it first;
        # this library then binds to the copies torch loaded (same SONAMEs).  Hosts without
        # torch (the C++ / Go bindings) are not affected.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        # RTLD_GLOBAL: libgogp_testhooks.so resolves the internal launchers against it
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, restype, argtypes in SYMBOLS:
            f = getattr(L, name)  # AttributeError if the symbol is not exported
            f.restype = restype
            f.argtypes = argtypes
        _lib = L
    return _lib
