"""TEST INFRASTRUCTURE ONLY -- Python access to the CPU oracle.

Two restatements of the reference algorithm (gp/gp.go, kernel/*.go) live here:

* ``Oracle``      -- ctypes binding of oracle/gogp_oracle.c, the *faithful*
                     restatement (pair loop, dense dK per parameter,
                     1/2 tr(aa^T dK - K^-1 dK) exactly as gp/gp.go:476-485).
                     O(P N^3); use for N up to a few hundred.
* ``FastOracle``  -- numpy/scipy twin of the same mathematics in the W-matrix
                     form  grad_p = 1/2 sum_ij (aa^T - K^-1)_ij dK_p,ij  with
                     LAPACK potrf/potri.  Used for N in the thousands and as
                     bench.py's ``cpu_baseline`` ("port").  It is validated
                     against ``Oracle`` and against the reference's known
                     answers in tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package gogp_amd never does.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

from gogp_amd.kernel import (CDesc, K_MATERN32, K_MATERN52, K_MATERN52_TEXTBOOK,
                             K_NORMAL, K_PERIODIC, NOISE_CONSTANT, NOISE_CONSTANT_PARAM, NOISE_UNIFORM,
                             SQRT3, SQRT5, build_desc)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgogp_oracle.so")

GOGP_OK, GOGP_EARG, GOGP_ENOTPD, GOGP_EHIP, GOGP_ESTATE = 0, 1, 2, 3, 4


def build(force: bool = False) -> str:
    """Compile the C oracle (gcc) if needed; returns the .so path."""
    src = os.path.join(_HERE, "gogp_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        dp = ctypes.POINTER(ctypes.c_double)
        i64 = ctypes.c_int64
        L.gogp_oracle_new.restype = ctypes.c_void_p
        L.gogp_oracle_new.argtypes = [ctypes.POINTER(CDesc)]
        L.gogp_oracle_free.argtypes = [ctypes.c_void_p]
        L.gogp_oracle_simil.restype = ctypes.c_double
        L.gogp_oracle_simil.argtypes = [ctypes.POINTER(CDesc), dp, dp, dp, dp]
        L.gogp_oracle_noise.restype = ctypes.c_double
        L.gogp_oracle_noise.argtypes = [ctypes.POINTER(CDesc), dp, dp, dp]
        L.gogp_oracle_absorb.argtypes = [ctypes.c_void_p, dp, dp, dp, dp, i64,
                                         ctypes.POINTER(i64)]
        L.gogp_oracle_lml.restype = ctypes.c_double
        L.gogp_oracle_lml.argtypes = [ctypes.c_void_p]
        L.gogp_oracle_observe.argtypes = [ctypes.c_void_p, dp, i64, dp, dp, i64, dp,
                                          ctypes.POINTER(i64)]
        L.gogp_oracle_gradient.argtypes = [ctypes.c_void_p, dp, i64]
        L.gogp_oracle_produce.argtypes = [ctypes.c_void_p, dp, i64, dp, dp]
        L.gogp_oracle_n.restype = i64
        L.gogp_oracle_n.argtypes = [ctypes.c_void_p]
        for name in ("alpha", "factor", "gram", "theta_simil", "theta_noise"):
            f = getattr(L, "gogp_oracle_" + name)
            f.restype = dp
            f.argtypes = [ctypes.c_void_p]
        L.gogp_oracle_dk.restype = dp
        L.gogp_oracle_dk.argtypes = [ctypes.c_void_p, i64]
        L.gogp_oracle_ndk.restype = i64
        L.gogp_oracle_ndk.argtypes = [ctypes.c_void_p]
        descp = ctypes.POINTER(CDesc)
        L.gogp_oracle_gram_omp.restype = None
        L.gogp_oracle_gram_omp.argtypes = [descp, dp, ctypes.c_double, dp, i64, dp]
        L.gogp_oracle_cross_omp.restype = None
        L.gogp_oracle_cross_omp.argtypes = [descp, dp, dp, i64, dp, i64, dp]
        L.gogp_oracle_grad_reduce_omp.restype = None
        L.gogp_oracle_grad_reduce_omp.argtypes = [descp, dp, dp, dp, dp, i64, dp]
        _lib = L
    return _lib


def _dp(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _arr(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        out = out.reshape(shape)
    return out


class NotPositiveDefinite(Exception):
    def __init__(self, pivot):
        super().__init__("Factorize: matrix is not positive definite (pivot %d)" % pivot)
        self.pivot = pivot


class Oracle:
    """Faithful restatement of gp.GP (C).  Method names follow the reference."""

    def __init__(self, ndim: int, simil, noise=None):
        self.desc = build_desc(ndim, simil, noise)
        self.ndim = ndim
        self.ns = self.desc.ntheta_simil
        self.nn = 0 if self.desc.noise_kind == NOISE_CONSTANT else 1
        self._h = lib().gogp_oracle_new(ctypes.byref(self.desc))
        self._X = np.zeros((0, ndim))
        self._y = np.zeros((0,))

    def __del__(self):
        try:
            if self._h:
                lib().gogp_oracle_free(self._h)
                self._h = None
        except Exception:
            pass

    # -- kernel evaluation ----------------------------------------------------
    def simil(self, theta, xa, xb, with_grad=False):
        theta, xa, xb = _arr(theta), _arr(xa), _arr(xb)
        g = np.zeros(self.ns + 2 * self.ndim) if with_grad else None
        v = lib().gogp_oracle_simil(ctypes.byref(self.desc), _dp(theta), _dp(xa), _dp(xb), _dp(g))
        return (v, g) if with_grad else v

    # -- gp.GP API --------------------------------------------------------------
    def set_data(self, X, y):
        self._X = _arr(X).reshape(-1, self.ndim)
        self._y = _arr(y).reshape(-1)
        assert len(self._X) == len(self._y)

    def Absorb(self, X, y, theta_simil, theta_noise=()):
        self.set_data(X, y)
        ts = _arr(theta_simil)
        tn = _arr(theta_noise) if self.nn else np.zeros(1)
        assert ts.size == self.ns
        piv = ctypes.c_int64(-1)
        rc = lib().gogp_oracle_absorb(self._h, _dp(ts), _dp(tn), _dp(self._X), _dp(self._y),
                                      len(self._y), ctypes.byref(piv))
        if rc == GOGP_ENOTPD:
            raise NotPositiveDefinite(piv.value)
        if rc != GOGP_OK:
            raise RuntimeError("oracle absorb rc=%d" % rc)

    def LML(self) -> float:
        return lib().gogp_oracle_lml(self._h)

    def Observe(self, x) -> float:
        """x as in gp/gp.go:366-373; the caller's array is left as the reference
        leaves it (exp/log round trip, <= 1 ulp drift)."""
        xa = _arr(x).copy()
        lml = ctypes.c_double(0.0)
        piv = ctypes.c_int64(-1)
        rc = lib().gogp_oracle_observe(self._h, _dp(xa), xa.size, _dp(self._X), _dp(self._y),
                                       len(self._y), ctypes.byref(lml), ctypes.byref(piv))
        if rc == GOGP_ENOTPD:
            raise NotPositiveDefinite(piv.value)
        if rc == GOGP_EARG:
            raise ValueError("len(x)")
        if rc != GOGP_OK:
            raise RuntimeError("oracle observe rc=%d" % rc)
        self._last_len = xa.size
        return lml.value

    def Gradient(self) -> np.ndarray:
        g = np.zeros(self._last_len)
        rc = lib().gogp_oracle_gradient(self._h, _dp(g), g.size)
        if rc != GOGP_OK:
            raise RuntimeError("oracle gradient rc=%d" % rc)
        return g

    def Produce(self, Z):
        Z = _arr(Z).reshape(-1, self.ndim)
        m = len(Z)
        mu, sigma = np.zeros(m), np.zeros(m)
        rc = lib().gogp_oracle_produce(self._h, _dp(Z), m, _dp(mu), _dp(sigma))
        if rc != GOGP_OK:
            raise RuntimeError("oracle produce rc=%d" % rc)
        return mu, sigma

    # -- cached state -------------------------------------------------------------
    @property
    def n(self) -> int:
        return lib().gogp_oracle_n(self._h)

    def _mat(self, getter, shape):
        p = getter(self._h)
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=shape).copy()

    @property
    def Alpha(self):
        return self._mat(lib().gogp_oracle_alpha, (self.n,))

    @property
    def L(self):
        return self._mat(lib().gogp_oracle_factor, (self.n, self.n))

    @property
    def K(self):
        return self._mat(lib().gogp_oracle_gram, (self.n, self.n))

    def dK(self, p):
        ptr = lib().gogp_oracle_dk(self._h, p)
        if not ptr:
            return None
        return np.ctypeslib.as_array(ptr, shape=(self.n, self.n)).copy()


# =============================================================================
# numpy/scipy twin (W-matrix form)
# =============================================================================

def _terms(desc: CDesc):
    return [desc.terms[i] for i in range(desc.nterms)]


def gram_np(desc: CDesc, theta_s: np.ndarray, A: np.ndarray, B: np.ndarray,
            want_grad: bool = False):
    """Similarity matrix k(A_i, B_j) and (optionally) the list of
    theta_p * dk/dtheta_p matrices (derivative w.r.t. log theta_p), following
    kernel/kernel.go with r^2 = sum_d ((a_d-b_d)/l_d)^2.  Differences are formed
    explicitly per dimension (no |a|^2+|b|^2-2ab cancellation)."""
    D = desc.ndim
    nA, nB = len(A), len(B)
    K = np.zeros((nA, nB))
    dK = [np.zeros((nA, nB)) for _ in range(desc.ntheta_simil)] if want_grad else None

    def diff(j):
        return A[:, j, None] - B[None, :, j]

    for T in _terms(desc):
        c = theta_s[T.scale_idx] if T.scale_idx >= 0 else 1.0
        ls = np.array([theta_s[T.len_idx + (j if T.ard else 0)] for j in range(D)])
        if T.kind == K_PERIODIC:
            p = T.period_mult * theta_s[T.period_idx]
            s2 = np.zeros((nA, nB))
            gp = np.zeros((nA, nB)) if want_grad else None
            for j in range(D):
                phi = (np.pi / p) * np.abs(diff(j))
                dd = np.sin(phi) / ls[j]
                s2 += dd * dd
                if want_grad:
                    gp += dd * np.cos(phi) * phi / ls[j]
            f = np.exp(-2 * s2)
            K += c * f
            if want_grad:
                if T.scale_idx >= 0:
                    dK[T.scale_idx] += c * f
                dK[T.period_idx] += c * f * 4 * gp
                if T.ard:
                    for j in range(D):
                        dd = np.sin((np.pi / p) * np.abs(diff(j))) / ls[j]
                        dK[T.len_idx + j] += c * f * 4 * dd * dd
                else:
                    dK[T.len_idx] += c * f * 4 * s2
            continue
        r2 = np.zeros((nA, nB))
        for j in range(D):
            u = diff(j)
            u *= 1.0 / ls[j]
            u *= u
            r2 += u
        if T.kind == K_NORMAL:
            f = np.exp(-0.5 * r2)
            dfdr2 = -0.5 * f
        else:
            r = np.sqrt(r2)
            if T.kind == K_MATERN32:
                e = np.exp(-SQRT3 * r)
                f = (1 + SQRT3 * r) * e
                dfdr2 = -1.5 * e
            elif T.kind == K_MATERN52:
                e = np.exp(-SQRT5 * r)
                f = (1 + SQRT5 * r + r2) * e
                dfdr2 = -0.5 * (3 + SQRT5 * r) * e
            elif T.kind == K_MATERN52_TEXTBOOK:
                e = np.exp(-SQRT5 * r)
                f = (1 + SQRT5 * r + (5.0 / 3.0) * r2) * e
                dfdr2 = -(5.0 / 6.0) * (1 + SQRT5 * r) * e
            else:
                raise ValueError("kind")
        K += c * f
        if want_grad:
            if T.scale_idx >= 0:
                dK[T.scale_idx] += c * f
            if T.ard:
                for j in range(D):
                    u = diff(j) / ls[j]
                    dK[T.len_idx + j] += c * dfdr2 * (-2.0) * u * u
            else:
                dK[T.len_idx] += c * dfdr2 * (-2.0) * r2
    return (K, dK) if want_grad else K


def potrf_blocked(K, nb=1024):
    """Lower Cholesky factor of the C-ordered matrix K, in place (the strict upper triangle is left as it
    was): right-looking blocked factorisation whose O(N^3) work is numpy matmul (threaded dgemm) on
    panel-by-block products.  LAPACK's threaded dpotrf (OpenBLAS) collapses at some sizes on hosts with a
    CPU quota far below the visible CPU count (GPU box, 16 of 256: N = 8192 ran at 32 GFLOP/s, as slow as
    N = 16384); the CPU baseline of bench.py takes whichever of the two is faster.  gonum's own Dpotrf is
    the same right-looking blocked algorithm (mat.Cholesky.Factorize, gp/gp.go:228)."""
    import scipy.linalg as sla
    n = K.shape[0]
    for k in range(0, n, nb):
        e = min(k + nb, n)
        try:
            L11 = sla.cholesky(K[k:e, k:e], lower=True, check_finite=False)
        except np.linalg.LinAlgError as ex:
            raise NotPositiveDefinite(-1) from ex
        K[k:e, k:e] = L11
        if e < n:
            L21 = sla.solve_triangular(L11, K[e:, k:e].T, lower=True, check_finite=False).T
            K[e:, k:e] = L21
            for j in range(e, n, nb):
                je = min(j + nb, n)
                K[j:, j:je] -= L21[j - e:, :] @ L21[j - e:je - e, :].T
    return K


def _tri_inv(L, nb):
    """Inverse of a lower triangular matrix by recursive halving: [[A, 0], [B, C]]^-1 =
    [[A^-1, 0], [-C^-1 B A^-1, C^-1]]; the products are plain (threaded) matmuls."""
    import scipy.linalg as sla
    n = L.shape[0]
    if n <= nb:
        Y, info = sla.lapack.dtrtri(np.asfortranarray(L), lower=1)
        assert info == 0
        return np.tril(Y)
    h = (n // 2 + nb - 1) // nb * nb
    Y = np.zeros((n, n))
    Y[:h, :h] = _tri_inv(L[:h, :h], nb)
    Y[h:, h:] = _tri_inv(L[h:, h:], nb)
    Y[h:, :h] = -(Y[h:, h:] @ (L[h:, :h] @ Y[:h, :h]))
    return Y


def potri_blocked(L, nb=1024):
    """K^-1 (lower triangle valid) from the lower Cholesky factor: Y = L^-1 by recursive halving, then
    K^-1 = Y^T Y block column by block column -- every O(N^3) product a threaded matmul (counterpart of
    potrf_blocked above; the reference forms K^-1 dK_p by SolveTo per parameter, gp/gp.go:480)."""
    n = L.shape[0]
    Y = _tri_inv(L, nb)
    Kinv = np.zeros((n, n))
    for i in range(0, n, nb):
        e = min(i + nb, n)
        # rows i:e of K^-1, columns 0:e = sum over k >= i of Y[k, i:e]^T Y[k, 0:e]
        Kinv[i:e, :e] = Y[i:, i:e].T @ Y[i:, :e]
    return Kinv


class FastOracle:
    """Restatement in the W-matrix form (hyperparameters-only Observe/Gradient,
    Absorb, Produce): LAPACK potrf/potri/potrs (scipy, OpenBLAS threads) for the
    O(N^3) parts; the O(N^2) pair loops either in C/OpenMP (``use_c=True``, the
    default: gogp_oracle_gram_omp / gogp_oracle_grad_reduce_omp) or in numpy
    (``use_c=False``, kept as an independent cross-check)."""

    def __init__(self, ndim: int, simil, noise=None, block: int = 1024, use_c: bool = True,
                 potrf: str = "lapack", potri: str = "lapack"):
        self.desc = build_desc(ndim, simil, noise)
        self.ndim = ndim
        self.ns = self.desc.ntheta_simil
        self.nn = 0 if self.desc.noise_kind == NOISE_CONSTANT else 1
        self.block = block
        self.use_c = use_c
        #: "lapack" (scipy dpotrf) or "blocked" (potrf_blocked above: dgemm-based)
        self.potrf = potrf
        self.potri = potri
        self.X = np.zeros((0, ndim))
        self.Y = np.zeros((0,))
        self.Lc = None
        self.Alpha = None
        #: seconds spent per phase since the last reset (bench.py's cpu_baseline reports them)
        self.timings = {}

    def _t(self, name, t0):
        import time
        self.timings[name] = self.timings.get(name, 0.0) + (time.perf_counter() - t0)

    def set_data(self, X, y):
        self.X = _arr(X).reshape(-1, self.ndim)
        self.Y = _arr(y).reshape(-1)

    def _noise_var(self, tn):
        if self.desc.noise_kind in (NOISE_CONSTANT, NOISE_CONSTANT_PARAM):
            return self.desc.noise_std ** 2
        return self.desc.noise_scale * tn[0] ** 2

    def _gram(self, ts, tn):
        n = len(self.X)
        K = np.empty((n, n))
        if self.use_c:
            lib().gogp_oracle_gram_omp(ctypes.byref(self.desc), _dp(ts), float(self._noise_var(tn)),
                                       _dp(self.X), n, _dp(K))
            return K
        b = self.block
        for i0 in range(0, n, b):
            K[i0:i0 + b] = gram_np(self.desc, ts, self.X[i0:i0 + b], self.X)
        K[np.diag_indices(n)] += self._noise_var(tn)
        return K

    def _factor(self, ts, tn):
        import scipy.linalg as sla
        self.ts, self.tn = _arr(ts), _arr(tn)
        n = len(self.X)
        if n == 0:
            self.Lc, self.Alpha = None, np.zeros(0)
            return
        import time
        t0 = time.perf_counter()
        K = self._gram(self.ts, self.tn)
        self._t("gram", t0)
        t0 = time.perf_counter()
        if self.potrf == "blocked":
            L = np.tril(potrf_blocked(K, self.block))
        else:
            try:
                L = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
            except np.linalg.LinAlgError as e:
                raise NotPositiveDefinite(-1) from e
        self._t("potrf", t0)
        t0 = time.perf_counter()
        self.Lc = L
        self.Alpha = sla.cho_solve((L, True), self.Y, check_finite=False)
        self._t("potrs", t0)

    def Absorb(self, X, y, theta_simil, theta_noise=()):
        self.set_data(X, y)
        self._factor(theta_simil, theta_noise)

    def LML(self) -> float:
        n = len(self.X)
        if n == 0:
            return 0.0
        return (-0.5 * n * math.log(2 * math.pi) - np.log(np.diag(self.Lc)).sum()
                - 0.5 * float(self.Y @ self.Alpha))

    def Observe(self, x) -> float:
        x = _arr(x)
        assert x.size == self.ns + self.nn, "FastOracle: hyperparameters-only form"
        th = np.exp(x)
        self._factor(th[:self.ns], th[self.ns:])
        return self.LML()

    def Gradient(self) -> np.ndarray:
        import scipy.linalg as sla
        n = len(self.X)
        P = self.ns + self.nn
        g = np.zeros(P)
        if n == 0:
            return g
        # K^-1 from the factor (dpotri).  LAPACK is column-major: handed the C-ordered factor as its
        # transpose (a Fortran-ordered VIEW, the upper factor U = L^T) it inverts without the two
        # 8 N^2-byte layout copies f2py would otherwise make; the result comes back as a Fortran-ordered
        # array whose UPPER triangle is valid, i.e. -- read as C order -- the lower triangle of K^-1
        import time
        t0 = time.perf_counter()
        if self.potri == "blocked":
            Kinv = potri_blocked(self.Lc, self.block)
        else:
            Kinv_f, info = sla.lapack.dpotri(self.Lc.T, lower=0, overwrite_c=0)
            assert info == 0
            Kinv = Kinv_f.T  # C-contiguous view, lower triangle valid
        self._t("potri", t0)
        a = np.ascontiguousarray(self.Alpha)
        if self.use_c:
            assert Kinv.flags.c_contiguous
            out = np.zeros(self.ns + 1)
            t0 = time.perf_counter()
            lib().gogp_oracle_grad_reduce_omp(ctypes.byref(self.desc), _dp(self.ts), _dp(self.X),
                                              _dp(a), _dp(Kinv), n, _dp(out))
            self._t("grad_reduce", t0)
            g[:self.ns] = out[:self.ns]
            trW = out[self.ns]
        else:
            Kinv = np.tril(Kinv) + np.tril(Kinv, -1).T
            b = self.block
            for i0 in range(0, n, b):
                W = np.outer(a[i0:i0 + b], a) - Kinv[i0:i0 + b]
                _, dK = gram_np(self.desc, self.ts, self.X[i0:i0 + b], self.X, want_grad=True)
                for p in range(self.ns):
                    g[p] += 0.5 * float((W * dK[p]).sum())
            trW = float(a @ a) - float(np.trace(Kinv))
        if self.nn and self.desc.noise_kind == NOISE_UNIFORM:
            g[self.ns] = 0.5 * trW * 2.0 * self.desc.noise_scale * self.tn[0] ** 2
        return g  # NOISE_CONSTANT_PARAM: K does not depend on the parameter, component 0

    def Produce(self, Z):
        import scipy.linalg as sla
        Z = _arr(Z).reshape(-1, self.ndim)
        m = len(Z)
        prior = np.array([gram_np(self.desc, self.ts, Z[i:i + 1], Z[i:i + 1])[0, 0]
                          for i in range(m)]) if m else np.zeros(0)
        if len(self.X) == 0:
            return np.zeros(m), np.sqrt(prior)
        if self.use_c:
            Ks = np.empty((len(self.X), m))
            lib().gogp_oracle_cross_omp(ctypes.byref(self.desc), _dp(self.ts), _dp(self.X),
                                        len(self.X), _dp(Z), m, _dp(Ks))
        else:
            Ks = gram_np(self.desc, self.ts, self.X, Z)  # n x m
        mu = Ks.T @ self.Alpha
        v = sla.cho_solve((self.Lc, True), Ks, check_finite=False)
        cov = np.einsum("ij,ij->j", Ks, v)
        with np.errstate(invalid="ignore"):
            sigma = np.sqrt(prior - cov)
        return mu, sigma
