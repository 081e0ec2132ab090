"""Host-side mirror of the reference's ``kernel`` package (kernel/kernel.go,
kernel/noise.go) plus the descriptor the device path needs.

The reference plugs kernels into ``gp.GP`` as Go values implementing

    type Kernel interface { Observe([]float64) float64; NTheta() int }   (gp/gp.go:14-17)

and calls ``Simil.Observe([theta | xa | xb])`` once per input pair
(gp/gp.go:110-111).  A GPU cannot call back into host code per pair, so every
kernel here ALSO carries a closed description (``terms``) from which
``build_desc`` makes the ``gogp_desc`` of include/gogp_hip.h.  ``Observe`` and
``NTheta`` keep the reference's names and argument layout, so user code and
tests written against the reference read the same.

Composition mirrors what the reference's tutorials do in Go source:
  * ``Scaled(k)``      -> ``x[0] * k.Observe(x[1:])``  (tutorial/barebones/kernel/kernel.go:14-18)
  * ``Sum(a, b)``      -> explicit index mapping       (tutorial/hyperpriors/kernel/kernel.go:12-25)
  * ``ScaledNoise(s)`` -> ``s * UniformNoise.Observe`` (tutorial/barebones/kernel/kernel.go:25-31)
  * ``ARD(k, ndim)``   -> one length scale per input dimension (build-defined,
                          SURVEY.md section 8d: the reference primitives are 1-D).
"""
from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

# ---- ctypes mirror of include/gogp_hip.h ------------------------------------
GOGP_MAX_TERMS = 4
GOGP_MAX_NDIM = 64

K_NORMAL, K_MATERN32, K_MATERN52, K_MATERN52_TEXTBOOK, K_PERIODIC = 0, 1, 2, 3, 4
NOISE_CONSTANT, NOISE_UNIFORM, NOISE_CONSTANT_PARAM = 0, 1, 2

SQRT3 = 1.7320508075688772  # kernel/kernel.go:51
SQRT5 = 2.2360679774997900  # kernel/kernel.go:52

#: default noise std when gp.GP.Noise is nil (gp/gp.go:43)
NONOISE = 1e-5


class CTerm(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("scale_idx", ctypes.c_int32),
        ("len_idx", ctypes.c_int32),
        ("ard", ctypes.c_int32),
        ("period_idx", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("period_mult", ctypes.c_double),
    ]


class CDesc(ctypes.Structure):
    _fields_ = [
        ("ndim", ctypes.c_int32),
        ("nterms", ctypes.c_int32),
        ("ntheta_simil", ctypes.c_int32),
        ("noise_kind", ctypes.c_int32),
        ("noise_std", ctypes.c_double),
        ("noise_scale", ctypes.c_double),
        ("terms", CTerm * GOGP_MAX_TERMS),
    ]


@dataclass
class Term:
    """One additive term  c * f(r)  of a similarity kernel."""

    kind: int
    scale_idx: int = -1
    len_idx: int = 0
    ard: bool = False
    period_idx: int = -1
    period_mult: float = 1.0

    def shifted(self, off: int) -> "Term":
        return Term(
            self.kind,
            self.scale_idx + off if self.scale_idx >= 0 else -1,
            self.len_idx + off,
            self.ard,
            self.period_idx + off if self.period_idx >= 0 else -1,
            self.period_mult,
        )


def _term_value(t: Term, theta: Sequence[float], xa: Sequence[float], xb: Sequence[float]) -> float:
    """Value of one term; the formulas of kernel/kernel.go with
    r^2 = sum_d ((xa_d-xb_d)/l_d)^2."""
    D = len(xa)
    c = theta[t.scale_idx] if t.scale_idx >= 0 else 1.0
    if t.kind == K_PERIODIC:  # kernel/kernel.go:44-47
        p = t.period_mult * theta[t.period_idx]
        s2 = 0.0
        for j in range(D):
            l = theta[t.len_idx + (j if t.ard else 0)]
            d = math.sin(math.pi * abs(xa[j] - xb[j]) / p) / l
            s2 += d * d
        return c * math.exp(-2 * s2)
    r2 = 0.0
    for j in range(D):
        l = theta[t.len_idx + (j if t.ard else 0)]
        u = (xa[j] - xb[j]) / l
        r2 += u * u
    if t.kind == K_NORMAL:  # kernel/kernel.go:23-26
        return c * math.exp(-r2 / 2)
    r = math.sqrt(r2)
    if t.kind == K_MATERN32:  # kernel/kernel.go:70-73
        return c * (1 + SQRT3 * r) * math.exp(-SQRT3 * r)
    if t.kind == K_MATERN52:  # kernel/kernel.go:89-92 (Go's 5/3 == 1)
        return c * (1 + SQRT5 * r + (5 // 3) * r * r) * math.exp(-SQRT5 * r)
    if t.kind == K_MATERN52_TEXTBOOK:
        return c * (1 + SQRT5 * r + (5.0 / 3.0) * r * r) * math.exp(-SQRT5 * r)
    raise ValueError("unknown kernel kind %r" % (t.kind,))


class SimilKernel:
    """A similarity kernel: the reference ``Kernel`` interface plus ``terms``."""

    def __init__(self, terms: List[Term], ntheta: int, name: str = "simil"):
        self.terms = terms
        self._ntheta = ntheta
        self.name = name

    # reference interface ---------------------------------------------------
    def NTheta(self) -> int:
        return self._ntheta

    def Observe(self, x: Sequence[float]) -> float:
        """x = [theta (NTheta) | xa (D) | xb (D)]  (gp/gp.go:173-175,110)."""
        nt = self._ntheta
        rest = len(x) - nt
        if rest <= 0 or rest % 2:
            raise ValueError("Observe: len(x)")
        D = rest // 2
        theta, xa, xb = x[:nt], x[nt:nt + D], x[nt + D:]
        return sum(_term_value(t, theta, xa, xb) for t in self.terms)

    def __repr__(self):
        return "<%s NTheta=%d>" % (self.name, self._ntheta)


def _single(kind: int, name: str) -> SimilKernel:
    return SimilKernel([Term(kind, -1, 0, False)], 1, name)


#: kernel.Normal (kernel/kernel.go:8-26): NTheta = 1 (length scale)
Normal = _single(K_NORMAL, "Normal")
#: kernel.Matern32 (kernel/kernel.go:55-73)
Matern32 = _single(K_MATERN32, "Matern32")
#: kernel.Matern52 (kernel/kernel.go:75-92), d^2 coefficient 1 as compiled by Go
Matern52 = _single(K_MATERN52, "Matern52")
#: textbook Matern-5/2 with coefficient 5/3 (not what the reference computes)
Matern52Textbook = _single(K_MATERN52_TEXTBOOK, "Matern52Textbook")
#: kernel.Periodic (kernel/kernel.go:28-47): theta = [l, p]
Periodic = SimilKernel([Term(K_PERIODIC, -1, 0, False, 1, 1.0)], 2, "Periodic")


def ARD(k: SimilKernel, ndim: int) -> SimilKernel:
    """Single-term kernel with one length scale per input dimension."""
    if len(k.terms) != 1 or k.terms[0].scale_idx >= 0:
        raise ValueError("ARD wraps a primitive kernel")
    t = k.terms[0]
    nt = ndim + (1 if t.kind == K_PERIODIC else 0)
    nt_term = Term(t.kind, -1, 0, True, ndim if t.kind == K_PERIODIC else -1, t.period_mult)
    return SimilKernel([nt_term], nt, "ARD(%s,%d)" % (k.name, ndim))


def Scaled(k: SimilKernel) -> SimilKernel:
    """``x[0] * k.Observe(x[1:])`` -- tutorial/barebones/kernel/kernel.go:14-18.
    theta = [c | theta of k]."""
    if len(k.terms) != 1 or k.terms[0].scale_idx >= 0:
        raise ValueError("Scaled wraps an unscaled single-term kernel")
    t = k.terms[0].shifted(1)
    t.scale_idx = 0
    return SimilKernel([t], k.NTheta() + 1, "Scaled(%s)" % k.name)


def Sum(parts: Sequence[SimilKernel], order: Optional[Sequence[int]] = None) -> SimilKernel:
    """Sum of kernels.  By default theta is the concatenation of the parts'
    parameter vectors; ``order`` (a permutation: new index of each old index)
    reproduces hand-written layouts such as [c1, c2, l1, l2, p] of
    tutorial/hyperpriors/kernel/kernel.go:12-25."""
    terms: List[Term] = []
    off = 0
    for k in parts:
        for t in k.terms:
            terms.append(t.shifted(off))
        off += k.NTheta()
    if len(terms) > GOGP_MAX_TERMS:
        raise ValueError("too many terms")
    if order is not None:
        if sorted(order) != list(range(off)):
            raise ValueError("order must be a permutation of range(NTheta)")
        for t in terms:
            if t.ard:
                raise ValueError("order with ARD terms is not supported")
            if t.scale_idx >= 0:
                t.scale_idx = order[t.scale_idx]
            t.len_idx = order[t.len_idx]
            if t.period_idx >= 0:
                t.period_idx = order[t.period_idx]
    return SimilKernel(terms, off, "Sum(%s)" % ",".join(k.name for k in parts))


def PeriodScaled(k: SimilKernel, mult: float) -> SimilKernel:
    """Periodic kernel whose period parameter is multiplied by a constant
    (the ``10*x[p]`` of tutorial/hyperpriors/kernel/kernel.go:24)."""
    terms = [Term(t.kind, t.scale_idx, t.len_idx, t.ard, t.period_idx, t.period_mult * mult)
             for t in k.terms]
    return SimilKernel(terms, k.NTheta(), "PeriodScaled(%s,%g)" % (k.name, mult))


class NoiseKernel:
    """A noise kernel: added to the diagonal only; args [theta_n | x]
    (gp/gp.go:133-135, kernel/noise.go)."""

    def __init__(self, kind: int, std: float = 0.0, scale: float = 1.0):
        self.kind = kind
        self.std = float(std)
        self.scale = float(scale)

    def NTheta(self) -> int:
        return 0 if self.kind == NOISE_CONSTANT else 1

    def Observe(self, x: Sequence[float]) -> float:
        if self.kind in (NOISE_CONSTANT, NOISE_CONSTANT_PARAM):  # kernel/noise.go:23-30
            return self.std * self.std
        return self.scale * x[0] * x[0]  # kernel/noise.go:43-49

    def __repr__(self):
        if self.kind == NOISE_CONSTANT:
            return "ConstantNoise(%g)" % self.std
        if self.kind == NOISE_CONSTANT_PARAM:
            return "ConstantNoiseParam(%g)" % self.std
        return "%g*UniformNoise" % self.scale


def ConstantNoise(std: float) -> NoiseKernel:
    """kernel.ConstantNoise(std): variance std^2, no parameters (kernel/noise.go:18-34)."""
    return NoiseKernel(NOISE_CONSTANT, std=std)


def ConstantNoiseParam(std: float) -> NoiseKernel:
    """Constant variance std^2 with ONE parameter the Gram matrix does not depend on -- the
    noise kernel of tutorial/anynoise/kernel/kernel.go:26-35 (``return 1e-5``, ``NTheta() == 1``):
    the parameter exists only so that the model's priors can use it; its LML gradient is 0."""
    return NoiseKernel(NOISE_CONSTANT_PARAM, std=std)


#: kernel.UniformNoise (kernel/noise.go:36-53): one parameter, the standard error
UniformNoise = NoiseKernel(NOISE_UNIFORM, scale=1.0)


def ScaledNoise(scale: float) -> NoiseKernel:
    """``scale * kernel.UniformNoise.Observe(x)`` -- tutorial/barebones/kernel/kernel.go:25-31."""
    return NoiseKernel(NOISE_UNIFORM, scale=scale)


def build_desc(ndim: int, simil: SimilKernel, noise: Optional[NoiseKernel]) -> CDesc:
    """Make the C descriptor for gp.GP{NDim, Simil, Noise}.  ``noise is None``
    means the reference default ConstantNoise(1e-5) (gp/gp.go:45-48)."""
    if not isinstance(simil, SimilKernel):
        raise TypeError(
            "Simil must be a gogp_amd.kernel.SimilKernel: the device path needs a "
            "closed kernel description (arbitrary host callables cannot run per pair on the GPU)")
    if noise is None:
        noise = ConstantNoise(NONOISE)
    if not isinstance(noise, NoiseKernel):
        raise TypeError("Noise must be a gogp_amd.kernel.NoiseKernel")
    if not (1 <= ndim <= GOGP_MAX_NDIM):
        raise ValueError("NDim out of range")
    if not (1 <= len(simil.terms) <= GOGP_MAX_TERMS):
        raise ValueError("number of terms out of range")
    d = CDesc()
    d.ndim = ndim
    d.nterms = len(simil.terms)
    d.ntheta_simil = simil.NTheta()
    d.noise_kind = noise.kind
    d.noise_std = noise.std
    d.noise_scale = noise.scale
    for i, t in enumerate(simil.terms):
        ct = d.terms[i]
        ct.kind = t.kind
        ct.scale_idx = t.scale_idx
        ct.len_idx = t.len_idx
        ct.ard = 1 if t.ard else 0
        ct.period_idx = t.period_idx
        ct.reserved = 0
        ct.period_mult = t.period_mult
        nlen = ndim if t.ard else 1
        if t.len_idx < 0 or t.len_idx + nlen > simil.NTheta():
            raise ValueError("length-scale index out of range (ARD kernel used with wrong NDim?)")
    return d
