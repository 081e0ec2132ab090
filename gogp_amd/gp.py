"""Host-side mirror of the reference's ``gp`` package (gp/gp.go, gp/model.go)
over the C ABI of include/gogp_hip.h.

``GP`` keeps the reference's field and method names -- ``NDim, Simil, Noise,
ThetaSimil, ThetaNoise, X, Y, Parallel`` and ``Absorb / LML / Produce / Observe
/ Gradient`` (gp/gp.go:20-38,80,244,258,374,418) -- with the same argument
meaning and the same error behaviour:

  * ``Absorb`` returns normally or raises ``FactorizeError`` where the Go method
    returns ``err`` (gp/gp.go:228-230);
  * ``Observe`` raises where the reference panics (gp/gp.go:398-405);
  * ``Produce`` returns ``(mu, sigma)`` or raises (gp/gp.go:338-340).

All arithmetic runs on the GPU through libgogp_hip.so; there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from .kernel import NoiseKernel, SimilKernel, build_desc


class GogpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("gogp_hip error %d: %s" % (code, msg))
        self.code = code


class FactorizeError(GogpError):
    """gp/gp.go:228-230: Factorize(K) failed, K is not positive definite."""

    def __init__(self, msg: str, pivot: int):
        super().__init__(_lib.GOGP_ENOTPD, msg)
        self.pivot = pivot


class ConditionError(GogpError):
    """gonum's mat.Condition error (cond > 1e16) that gp/gp.go:233-236 returns from Absorb
    and turns into a panic in Observe: K factored, but it is numerically singular.  The state
    (L, Alpha, LML) was stored before the error was reported, as in gonum."""

    def __init__(self, msg: str):
        super().__init__(_lib.GOGP_ECOND, msg)


def _dp(a: Optional[np.ndarray]):
    if a is None:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _arr(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class GP:
    """Type GP is the barebone implementation of GP (gp/gp.go:19-38)."""

    def __init__(self, NDim: int, Simil: SimilKernel, Noise: Optional[NoiseKernel] = None,
                 ThetaSimil: Optional[Sequence[float]] = None,
                 ThetaNoise: Optional[Sequence[float]] = None,
                 X=None, Y=None, Parallel: bool = False, device: int = -1, precision: int = 64):
        self.NDim = int(NDim)
        self.Simil = Simil
        self.Noise = Noise
        # gp/gp.go:45-57 defaults(): Noise nil => ConstantNoise(1e-5) (done by
        # build_desc); zero theta vectors when empty
        self._desc = build_desc(self.NDim, Simil, Noise)
        self._ns = Simil.NTheta()
        self._nn = 0 if self._desc.noise_kind == 0 else 1
        self.ThetaSimil: List[float] = list(ThetaSimil) if ThetaSimil is not None else [0.0] * self._ns
        self.ThetaNoise: List[float] = list(ThetaNoise) if ThetaNoise is not None else [0.0] * self._nn
        #: accepted for source compatibility; the device path is always parallel
        self.Parallel = Parallel
        #: HIP device index of the handle (-1: the device current at construction)
        self.device = int(device)
        self._h = ctypes.c_void_p()
        L = _lib.lib()
        rc = L.gogp_create(ctypes.byref(self._desc), int(device), ctypes.byref(self._h))
        if rc != _lib.GOGP_OK:
            msg = L.gogp_last_error(None).decode()
            self._h = ctypes.c_void_p()
            raise GogpError(rc, msg)
        #: 64: fp64 throughout (the reference's arithmetic).  32: the N x N matrices and the
        #: O(N^3) products in fp32 (BASELINE configs[4]); inputs, kernel evaluation, diagonal
        #: blocks, vectors and reductions stay fp64 -- include/gogp_hip.h, option "precision"
        self.precision = int(precision)
        if self.precision != 64:
            self._check(L.gogp_set_option(self._h, b"precision", self.precision))
        self._X = np.zeros((0, self.NDim))
        self._Y = np.zeros((0,))
        self._data_dirty = True
        self._with_obs = False
        self._last_len = self._ns + self._nn
        if X is not None:
            self.X = X
            self.Y = Y if Y is not None else []

    # ---- plumbing ------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().gogp_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc == _lib.GOGP_OK:
            return
        L = _lib.lib()
        msg = L.gogp_last_error(self._h).decode()
        if rc == _lib.GOGP_ENOTPD:
            raise FactorizeError(msg, int(L.gogp_notpd_index(self._h)))
        if rc == _lib.GOGP_ECOND:
            raise ConditionError(msg)
        raise GogpError(rc, msg)

    # ---- data: gp.GP.X / gp.GP.Y (gp/gp.go:27-28) --------------------------------
    @property
    def X(self) -> np.ndarray:
        return self._X

    @X.setter
    def X(self, x):
        self._X = _arr(x).reshape(-1, self.NDim)
        self._data_dirty = True

    @property
    def Y(self) -> np.ndarray:
        return self._Y

    @Y.setter
    def Y(self, y):
        self._Y = _arr(y).reshape(-1)
        self._data_dirty = True

    def _push_data(self):
        if not self._data_dirty:
            return
        if len(self._X) != len(self._Y):
            raise ValueError("len(X) != len(Y)")
        self._check(_lib.lib().gogp_set_data(self._h, _dp(self._X), _dp(self._Y), len(self._Y)))
        self._data_dirty = False

    def set_data_device(self, dX_ptr: int, dy_ptr: int, n: int):
        """Inputs already resident in HBM (device pointers, e.g. torch .data_ptr())."""
        self._check(_lib.lib().gogp_set_data_device(self._h, ctypes.c_void_p(dX_ptr),
                                                    ctypes.c_void_p(dy_ptr), int(n)))
        self._data_dirty = False

    # ---- gp.GP.Absorb (gp/gp.go:80-87) -----------------------------------------------
    def Absorb(self, x, y) -> None:
        """Absorb absorbs observations into the process.  Parameters are taken
        from ThetaSimil / ThetaNoise (natural scale, gp/gp_test.go:29)."""
        self.X, self.Y = x, y
        self._push_data()
        ts = _arr(self.ThetaSimil)
        tn = _arr(self.ThetaNoise) if self._nn else np.zeros(1)
        if ts.size != self._ns:
            raise ValueError("len(ThetaSimil)")
        self._with_obs = False
        self._check(_lib.lib().gogp_absorb(self._h, _dp(ts), _dp(tn)))

    # ---- gp.GP.LML (gp/gp.go:244-253) -----------------------------------------------
    def LML(self) -> float:
        v = ctypes.c_double(0.0)
        self._check(_lib.lib().gogp_lml(self._h, ctypes.byref(v)))
        return v.value

    # ---- gp.GP.Produce (gp/gp.go:258-360) --------------------------------------------
    def Produce(self, x):
        z = _arr(x).reshape(-1, self.NDim)
        m = len(z)
        mu, sigma = np.zeros(m), np.zeros(m)
        if m:
            if self._data_dirty and len(self._Y) == 0:
                self._push_data()
            self._check(_lib.lib().gogp_produce(self._h, _dp(z), m, _dp(mu), _dp(sigma)))
        return mu, sigma

    # ---- gp.GP.Observe (gp/gp.go:374-413) ---------------------------------------------
    def Observe(self, x) -> float:
        """x = log-transformed hyperparameters [| inputs | outputs].  Raises where
        the reference panics.  x itself is not modified (the reference's in-place
        exp/log round trip, gp/gp.go:378-381,408-410, is not reproduced)."""
        xa = _arr(x).reshape(-1)
        P = self._ns + self._nn
        if xa.size < P:
            raise ValueError("len(x)")
        lml = ctypes.c_double(0.0)
        L = _lib.lib()
        if xa.size == P:
            self._push_data()
            self._with_obs = False
            rc = L.gogp_observe(self._h, _dp(xa), xa.size, ctypes.byref(lml))
        else:
            rest = xa.size - P
            n = rest // (self.NDim + 1)
            if n * (self.NDim + 1) != rest:
                raise ValueError("len(x)")  # gp/gp.go:398-400 panic("len(x)")
            rc = L.gogp_observe_full(self._h, _dp(xa), xa.size, ctypes.byref(lml))
            # gp/gp.go:391-396: X, Y are re-sliced from x -- also when the factorisation then
            # fails (the reference panics after the assignment): the device holds these data now
            self._X = xa[P:P + n * self.NDim].reshape(n, self.NDim).copy()
            self._Y = xa[P + n * self.NDim:].copy()
            self._data_dirty = rc != _lib.GOGP_OK  # after a failure: re-upload before the next call
            self._with_obs = rc == _lib.GOGP_OK
        self._check(rc)
        theta = np.exp(xa[:P])
        self.ThetaSimil = list(theta[:self._ns])  # gp/gp.go:384-385
        self.ThetaNoise = list(theta[self._ns:])
        self._last_len = xa.size
        return lml.value

    # ---- gp.GP.Gradient (gp/gp.go:418-499) ---------------------------------------------
    def Gradient(self) -> np.ndarray:
        g = np.zeros(self._last_len)
        self._check(_lib.lib().gogp_gradient(self._h, _dp(g), g.size))
        return g

    # ---- cached computations: gp.GP.L, gp.GP.Alpha (gp/gp.go:34-37) ----------------------
    @property
    def Alpha(self) -> np.ndarray:
        n = int(_lib.lib().gogp_n(self._h))
        a = np.zeros(n)
        if n:
            self._check(_lib.lib().gogp_get_alpha(self._h, _dp(a)))
        return a

    @property
    def L(self) -> np.ndarray:
        """Lower Cholesky factor (gonum's mat.Cholesky holds U = L^T)."""
        n = int(_lib.lib().gogp_n(self._h))
        out = np.zeros((n, n))
        if n:
            self._check(_lib.lib().gogp_get_factor(self._h, _dp(out)))
        return out

    def L_rows(self, rows) -> np.ndarray:
        """Selected rows of L (len(rows) x n), for checks at sizes where the whole factor
        is not wanted on the host."""
        n = int(_lib.lib().gogp_n(self._h))
        idx = np.ascontiguousarray(np.asarray(rows, dtype=np.int64))
        out = np.zeros((idx.size, n))
        if n and idx.size:
            self._check(_lib.lib().gogp_get_factor_rows(
                self._h, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), idx.size, _dp(out)))
        return out

    def L_diag(self) -> np.ndarray:
        n = int(_lib.lib().gogp_n(self._h))
        d = np.zeros(n)
        if n:
            self._check(_lib.lib().gogp_get_factor_diag(self._h, _dp(d)))
        return d

    def restore(self, L, Alpha) -> None:
        """Produce on stored results (gp/gp.go:255-257): re-install ThetaSimil,
        ThetaNoise, X, L, Alpha without refactorising."""
        self._push_data()
        ts = _arr(self.ThetaSimil)
        tn = _arr(self.ThetaNoise) if self._nn else np.zeros(1)
        Lm, al = _arr(L), _arr(Alpha)
        self._check(_lib.lib().gogp_set_factor(self._h, _dp(ts), _dp(tn), _dp(Lm), _dp(al)))

    # ---- measurement hooks ----------------------------------------------------------------
    def graph_info(self):
        """(nodes of the candidates' launch graph in use, whether the runtime refused an explicitly built graph)."""
        nodes, refused = ctypes.c_int64(0), ctypes.c_int(0)
        self._check(_lib.lib().gogp_graph_info(self._h, ctypes.byref(nodes), ctypes.byref(refused)))
        return int(nodes.value), bool(refused.value)

    def set_option(self, name: str, value: int):
        self._check(_lib.lib().gogp_set_option(self._h, name.encode(), int(value)))

    def profile_enable(self, on: bool = True):
        self._check(_lib.lib().gogp_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """(sum of launch durations ms, launches, launched flops, union busy ms)"""
        ms, nl, fl, bz = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        self._check(_lib.lib().gogp_profile_read(self._h, ctypes.byref(ms), ctypes.byref(nl),
                                                 ctypes.byref(fl), ctypes.byref(bz)))
        return ms.value, nl.value, fl.value, bz.value

    def observe_gradient_candidates(self, xs, strict=True):
        """LML and gradient of k candidate parameter vectors (rows of xs, log theta) on this GP's
        data in ONE launch sequence (gogp_observe_gradient_candidates): what Observe(xs[c]) +
        Gradient() would return for each c, without touching the GP's own state (a ShardedGP evaluates
        them one after the other in its own tiles and afterwards holds the last one's factorisation).  Counterpart:
        candidates evaluated concurrently by the reference's optimiser (optimize.Settings.
        Concurrent, tutorial/tutorial.go:30,141).  Returns (lmls[k], grads[k x P], status[k]):
        a candidate whose matrix is not positive definite has status GOGP_ENOTPD, lml NaN and a
        zero gradient instead of raising (the other candidates are still valid).  strict=False: a candidate
        with unusable parameters (status GOGP_EARG, e.g. exp(x) overflows) is reported the same way instead of
        raising -- the C ABI's per-candidate contract, on one GPU and on the shards alike."""
        xs = _arr(xs)
        P = self._ns + self._nn
        xs = xs.reshape(-1, P) if P else xs.reshape(len(xs), 0)
        k = xs.shape[0]
        self._push_data()
        lmls, grads = np.zeros(k), np.zeros((k, P))
        st = (ctypes.c_int * k)()
        rc = _lib.lib().gogp_observe_gradient_candidates(self._h, k, _dp(xs), P, _dp(lmls), _dp(grads), st)
        status = np.array(list(st), dtype=int)
        soft = (_lib.GOGP_OK, _lib.GOGP_ENOTPD, _lib.GOGP_ECOND) + (() if strict else (_lib.GOGP_EARG,))
        if rc not in soft or any(int(v) not in soft for v in status):
            self._check(rc if rc not in soft else _lib.GOGP_EARG)
        return lmls, grads, status

    def profile_read_launches(self):
        """Per launch of the tile kernel since profile_enable(True): arrays (start ms, end ms, flops, tag);
        tag = mode * 1e8 + (K / 16) * 1e5 + tiles.  Call before profile_read (which resets)."""
        L = _lib.lib()
        n = ctypes.c_int64(0)
        self._check(L.gogp_profile_read_launches(self._h, 0, None, None, None, None, ctypes.byref(n)))
        k = int(n.value)
        t0, t1, fl = np.zeros(k), np.zeros(k), np.zeros(k)
        tag = np.zeros(k, dtype=np.int64)
        if k:
            self._check(L.gogp_profile_read_launches(self._h, k, _dp(t0), _dp(t1), _dp(fl),
                                                     tag.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                     ctypes.byref(n)))
        return t0, t1, fl, tag

    def profile_read_aux(self, cls: int):
        """(sum of durations ms, timed launch groups) of one O(N^2) kernel class:
        0 Gram build, 1 gradient reduction, 2 cross-covariance (Produce)."""
        ms, nl = ctypes.c_double(0), ctypes.c_int64(0)
        self._check(_lib.lib().gogp_profile_read_aux(self._h, int(cls), ctypes.byref(ms), ctypes.byref(nl)))
        return ms.value, nl.value


def observe_gradient_batch(gps: Sequence[GP], xs) -> tuple:
    """Observe(xs[i]) + Gradient() on gps[i] for all i at once (hyperparameters-only form;
    every GP holds its own copy of the data): the k evaluations overlap on the GPU.
    Counterpart: candidates evaluated concurrently by the reference's optimiser
    (optimize.Settings.Concurrent = NTASKS, tutorial/tutorial.go:30,141).
    Returns (lmls[k], grads[k x P])."""
    k = len(gps)
    xs = _arr(xs).reshape(k, -1)
    P = xs.shape[1]
    for g in gps:
        if P != g._ns + g._nn:
            raise ValueError("len(x)")
        g._push_data()
        g._with_obs = False
    hs = (ctypes.c_void_p * k)(*[g._h for g in gps])
    lmls, grads = np.zeros(k), np.zeros((k, P))
    st = (ctypes.c_int * k)()
    rc = _lib.lib().gogp_observe_gradient_batch(hs, k, _dp(xs), P, _dp(lmls), _dp(grads), st)
    for i, g in enumerate(gps):
        if st[i] != _lib.GOGP_OK:
            g._check(st[i])
        theta = np.exp(xs[i])
        g.ThetaSimil, g.ThetaNoise = list(theta[:g._ns]), list(theta[g._ns:])
        g._last_len = P
    if rc != _lib.GOGP_OK:
        raise GogpError(rc, "observe_gradient_batch")
    return lmls, grads


class Model:
    """gp.Model (gp/model.go:9-28): GP plus priors on the hyperparameters.
    ``Priors`` is any object with Observe(x) -> float and Gradient() -> array."""

    def __init__(self, gp: GP, Priors):
        self.GP = gp
        self.Priors = Priors
        self._gGrad = None
        self._pGrad = None

    def Observe(self, x) -> float:
        gll = self.GP.Observe(x)
        self._gGrad = self.GP.Gradient()
        pll = self.Priors.Observe(x)
        self._pGrad = np.asarray(self.Priors.Gradient(), dtype=float)
        return gll + pll

    def Gradient(self) -> np.ndarray:
        g = self._gGrad.copy()
        g[:len(self._pGrad)] += self._pGrad
        return g


def mfma_f64_peak(iters: int = 20000, device: int = -1, details: bool = False):
    """fp64 MFMA issue-rate microbenchmark used to calibrate the roofline:
    TFLOP/s, or (TFLOP/s, cycles per MFMA on one SIMD, shader clock MHz)."""
    v, c, m = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    rc = _lib.hooks().gogp_mfma_f64_peak(device, iters, ctypes.byref(v), ctypes.byref(c),
                                       ctypes.byref(m))
    if rc != _lib.GOGP_OK:
        raise GogpError(rc, "mfma_f64_peak")
    return (v.value, c.value, m.value) if details else v.value


def mfma_f32_peak(iters: int = 20000, device: int = -1, details: bool = False):
    """The same microbenchmark for the fp32 path's v_mfma_f32_32x32x2_f32."""
    v, c, m = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    rc = _lib.hooks().gogp_mfma_f32_peak(device, iters, ctypes.byref(v), ctypes.byref(c), ctypes.byref(m))
    if rc != _lib.GOGP_OK:
        raise GogpError(rc, "mfma_f32_peak")
    return (v.value, c.value, m.value) if details else v.value


def dgemm_nt_check(A: np.ndarray, B: np.ndarray, C: np.ndarray, alpha=1.0, beta=0.0,
                  device: int = -1) -> np.ndarray:
    """C = beta*C + alpha*A@B.T on the GPU tile kernel (test hook)."""
    A, B = _arr(A), _arr(B)
    out = _arr(C).copy()
    M, K = A.shape
    N = B.shape[0]
    rc = _lib.hooks().gogp_test_dgemm_nt(device, M, N, K, alpha, _dp(A), _dp(B), beta, _dp(out))
    if rc != _lib.GOGP_OK:
        raise GogpError(rc, "test_dgemm_nt")
    return out


def diag256_check(A: np.ndarray, device: int = -1):
    """Factor + invert one 256x256 SPD block on the diagonal-block kernel (test /
    diagnostic hook): returns (L, Linv, stamps[24], elapsed_us)."""
    A = _arr(A)
    assert A.shape == (256, 256)
    L, X = np.zeros((256, 256)), np.zeros((256, 256))
    st = (ctypes.c_uint64 * 32)()
    us = ctypes.c_double(0.0)
    rc = _lib.hooks().gogp_test_diag256(device, _dp(A), _dp(L), _dp(X), st, ctypes.byref(us))
    if rc != _lib.GOGP_OK:
        raise GogpError(rc, "test_diag256")
    return L, X, np.array(list(st), dtype=np.uint64), us.value


def bench_gemm(mode: int, mt: int, nt: int, K: int, reps: int = 5, device: int = -1):
    """Time the tile kernel on one shape: returns (ms per launch, TFLOP/s)."""
    ms, tf = ctypes.c_double(0.0), ctypes.c_double(0.0)
    rc = _lib.hooks().gogp_bench_gemm(device, mode, mt, nt, K, reps, ctypes.byref(ms), ctypes.byref(tf))
    if rc != _lib.GOGP_OK:
        raise GogpError(rc, "bench_gemm")
    return ms.value, tf.value
