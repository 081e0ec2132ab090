"""The tutorial forecast harness (gogp_amd/tutorial.py; tutorial/tutorial.go:56-272).

CPU: load() and the harness logic driven by the oracle behind a GP-shaped adapter.
GPU: the same run on the HIP path must write the same forecasts as the oracle-backed run
(barebones case study: c*Matern32 + 0.01*sigma^2, tutorial/barebones/kernel/kernel.go:14-31,
on the reference's 20-row data file, committed as tests/golden/barebones.csv).
"""
import io
import os

import numpy as np
import pytest

from gogp_amd import kernel, tutorial
from oracle.oracle import Oracle

SIMIL = kernel.Scaled(kernel.Matern32)
NOISE = kernel.ScaledNoise(0.01)


class OracleGP:
    """The faithful CPU oracle with gp.GP's field/method shape (test-side only)."""

    def __init__(self, ndim, simil, noise):
        self.o = Oracle(ndim, simil, noise)
        self.NDim = ndim
        self.X = np.zeros((0, ndim))
        self.Y = np.zeros(0)
        self.Parallel = False
        self._P = self.o.ns + self.o.nn

    def Observe(self, x):
        x = np.asarray(x, dtype=float)
        if x.size == self._P:
            self.o.set_data(self.X, self.Y)
        else:  # gp/gp.go:391-396: X, Y are re-sliced out of x
            n = (x.size - self._P) // (self.NDim + 1)
            self.X = x[self._P:self._P + n * self.NDim].reshape(n, self.NDim).copy()
            self.Y = x[self._P + n * self.NDim:].copy()
            self.o.set_data(np.zeros((0, self.NDim)), np.zeros(0))
        return self.o.Observe(x)

    def Gradient(self):
        return self.o.Gradient()

    def Produce(self, Z):
        return self.o.Produce(Z)


@pytest.fixture()
def knobs():
    saved = {k: getattr(tutorial, k) for k in
             ("OPTINP", "MINOPT", "ALG", "ITERS", "THRESHOLD", "RATE", "NONORMALIZE", "OUTOFSAMPLE", "SEED", "NTASKS")}
    yield tutorial
    for k, v in saved.items():
        setattr(tutorial, k, v)


DEFAULTS = dict(OPTINP=False, MINOPT=0, ALG="lbfgs", ITERS=1000, THRESHOLD=1e-6, RATE=0.01,
                NONORMALIZE=False, OUTOFSAMPLE=False, SEED=None, NTASKS=0)


def _run(gp, golden_dir, model=None, data="barebones.csv", ntheta=3, **kn):
    for k, v in dict(DEFAULTS, **kn).items():
        setattr(tutorial, k, v)
    out = io.StringIO()
    with open(os.path.join(golden_dir, data)) as f:
        tutorial.Evaluate(gp, model if model is not None else gp, np.zeros(ntheta), f, out, log=io.StringIO())
    return [[float(v) for v in line.split(",")] for line in out.getvalue().strip().split("\n")], out.getvalue()


def test_load(golden_dir):
    with open(os.path.join(golden_dir, "barebones.csv")) as f:
        X, y = tutorial.load(f)
    assert X.shape == (20, 1) and y.shape == (20,)
    assert X[1, 0] == 0.3141592653589793 and y[0] == -0.04322589452340684
    X2, y2 = tutorial.load(io.StringIO("1,2,3\n4,5,6\n"))
    np.testing.assert_array_equal(X2, [[1, 2], [4, 5]])
    np.testing.assert_array_equal(y2, [3, 6])
    with pytest.raises(ValueError):
        tutorial.load(io.StringIO("1,x\n"))
    X0, y0 = tutorial.load(io.StringIO(""))
    assert len(X0) == 0 and len(y0) == 0


def test_format_like_go():
    assert tutorial._f(1.5) == "1.500000" and tutorial._f(float("nan")) == "NaN"
    assert tutorial._f(float("inf")) == "+Inf" and tutorial._f(-1e-9) == "-0.000000"


def test_evaluate_on_oracle(knobs, golden_dir):
    rows, text = _run(OracleGP(1, SIMIL, NOISE), golden_dir, SEED=3, ITERS=40, OUTOFSAMPLE=True)
    assert len(rows) == 20 + 19  # one forecast per point + out-of-sample tail (tutorial.go:198-225)
    first = rows[0]
    # columns: z, y, mu, sigma, lml0, lml, theta...  (tutorial.go:184-195); no data yet => LML 0,
    # prior forecast
    assert len(first) == 1 + 5 + 3 and first[4] == 0.0 and first[5] == 0.0
    with open(os.path.join(golden_dir, "barebones.csv")) as f:
        X, y = tutorial.load(f)
    for r, xi, yi in zip(rows[:20], X, y):
        assert abs(r[0] - xi[0]) < 1e-6 and abs(r[1] - yi) < 1e-6  # y is de-normalised again
    for r in rows[2:20]:
        assert r[5] >= r[4] - 1e-6  # optimisation never lowers the LML
    # the fit gets useful: late one-step forecasts land near the truth
    err = np.array([abs(r[2] - r[1]) for r in rows[10:20]])
    assert np.median(err) < 0.3
    assert "nan," in text.split("\n")[20]  # out-of-sample rows carry 'nan' for y
    assert len(rows[20]) == 1 + 3


def test_evaluate_adam_and_optinp_on_oracle(knobs, golden_dir):
    rows, _ = _run(OracleGP(1, SIMIL, NOISE), golden_dir, SEED=4, ALG="adam", ITERS=5, RATE=0.05)
    assert len(rows) == 20 and all(np.isfinite(r[5]) for r in rows)
    rows, _ = _run(OracleGP(1, SIMIL, NOISE), golden_dir, SEED=5, OPTINP=True, ITERS=3)
    assert len(rows) == 20 and all(np.isfinite(r[5]) for r in rows)


@pytest.mark.gpu
def test_evaluate_hip_matches_oracle(knobs, golden_dir):
    from gogp_amd import gp as G
    # no optimisation (MINOPT above N): rows are pure Observe + Produce => tight agreement
    want, _ = _run(OracleGP(1, SIMIL, NOISE), golden_dir, SEED=11, MINOPT=100, OUTOFSAMPLE=True)
    got, _ = _run(G.GP(1, SIMIL, NOISE), golden_dir, SEED=11, MINOPT=100, OUTOFSAMPLE=True)
    assert len(got) == len(want) == 39
    for g, w in zip(got, want):
        np.testing.assert_allclose(g, w, rtol=0, atol=2e-6)  # %f prints 6 decimals
    # optimised run, same seed: same forecasts (L-BFGS follows the same path while LML and
    # gradient agree to ~1e-13; allow for late divergence of the iterates)
    want, _ = _run(OracleGP(1, SIMIL, NOISE), golden_dir, SEED=12, ITERS=30)
    got, _ = _run(G.GP(1, SIMIL, NOISE), golden_dir, SEED=12, ITERS=30)
    for g, w in zip(got, want):
        assert abs(g[5] - w[5]) <= 1e-3 * max(1.0, abs(w[5])), (g, w)   # final LML
        assert abs(g[2] - w[2]) <= 1e-3 * max(1.0, abs(w[2])), (g, w)   # forecast mean
    # NTASKS = 4 (optimize.Settings.Concurrent, tutorial.go:141): four trial steps per launch sequence,
    # the identical optimisation path -> the identical output text
    seq, seq_text = _run(G.GP(1, SIMIL, NOISE), golden_dir, SEED=12, ITERS=30)
    par, par_text = _run(G.GP(1, SIMIL, NOISE), golden_dir, SEED=12, ITERS=30, NTASKS=4)
    assert par_text == seq_text
    # full form (inputs and outputs in x), a few steps
    want, _ = _run(OracleGP(1, SIMIL, NOISE), golden_dir, SEED=13, OPTINP=True, ITERS=2)
    got, _ = _run(G.GP(1, SIMIL, NOISE), golden_dir, SEED=13, OPTINP=True, ITERS=2)
    for g, w in zip(got, want):
        assert abs(g[5] - w[5]) <= 1e-5 * max(1.0, abs(w[5])), (g, w)


# ---- the reference's other case studies: priors through gp.Model (gp/model.go:9-28) ------------------
def _fd(pri_factory, x, idx):
    out = []
    for i in idx:
        h = 1e-6
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        a, b = pri_factory(), pri_factory()
        if hasattr(a, "Y"):  # the latent-output model memoises the outputs of its FIRST call
            a.Observe(x), b.Observe(x)
        out.append((a.Observe(xp) - b.Observe(xm)) / (2 * h))
    return np.array(out)


def test_case_study_priors_gradients():
    """tutorial/hyperpriors/model/model.go and tutorial/anynoise/model/model.go restated with
    hand-written gradients (the reference differentiates them with infergo's tape)."""
    from gogp_amd import priors
    rng = np.random.default_rng(5)
    x = rng.normal(size=6)
    p = priors.HyperPriors()
    v = p.Observe(x)
    np.testing.assert_allclose(p.Gradient(), _fd(priors.HyperPriors, x, range(6)), rtol=1e-6, atol=1e-8)
    # Normal.Logp(-1, 1, x[c1]) + ... at x = 0: closed form
    z = priors.HyperPriors().Observe(np.zeros(6))
    want = (-0.5 - 0.5 * np.log(2.0) ** 2 - 2 * np.log(2.0)) - 6 * 0.5 * np.log(2 * np.pi)
    assert abs(z - want) < 1e-12 and np.isfinite(v)
    n = 5
    xa = np.concatenate([rng.normal(size=3), rng.uniform(0, 1, n), rng.normal(size=n)])
    q = priors.AnyNoisePriors()
    q.Observe(xa)                      # memoises the outputs
    xb = xa.copy()
    xb[3 + n:] += 0.3 * rng.normal(size=n)
    q.Observe(xb)
    idx = [0, 1, 2] + list(range(3 + n, 3 + 2 * n))

    def factory():
        r = priors.AnyNoisePriors()
        r.Observe(xa)
        return r

    fd = []
    for i in idx:
        h = 1e-6
        xp, xm = xb.copy(), xb.copy()
        xp[i] += h
        xm[i] -= h
        fd.append((factory().Observe(xp) - factory().Observe(xm)) / (2 * h))
    np.testing.assert_allclose(q.Gradient()[idx], fd, rtol=1e-5, atol=1e-7)
    assert not q.Gradient()[3:3 + n].any()  # the priors do not depend on the inputs


HYPER_SIMIL = kernel.Sum([kernel.Scaled(kernel.Matern52), kernel.Scaled(kernel.PeriodScaled(kernel.Periodic, 10.0))],
                         order=[0, 2, 1, 3, 4])  # [c1, c2, l1, l2, p]: tutorial/hyperpriors/kernel/kernel.go:12-25


def test_hyperpriors_case_study_on_oracle(knobs, golden_dir):
    from gogp_amd import gp as G
    from gogp_amd import priors
    o = OracleGP(1, HYPER_SIMIL, kernel.ScaledNoise(0.01))
    rows, _ = _run(o, golden_dir, model=G.Model(o, priors.HyperPriors()), data="hyperpriors.csv", ntheta=6,
                   SEED=7, ITERS=15)
    assert len(rows) == 44 and all(len(r) == 1 + 5 + 6 for r in rows)
    assert all(r[5] >= r[4] - 1e-6 for r in rows[2:])  # optimisation never lowers LML + log prior
    err = np.array([abs(r[2] - r[1]) for r in rows[24:]])
    assert np.median(err) < 0.5  # trend + seasonality forecasts land near the data


@pytest.mark.gpu
def test_case_studies_hip_match_oracle(knobs, golden_dir):
    """hyperpriors (trend + seasonality with priors) and anynoise (latent outputs, Laplacian noise,
    full Observe form, inputs' gradient wiped: tutorial/anynoise/main.go:29-47) through the forecast
    harness: the HIP path writes the forecasts of the oracle-backed run."""
    from gogp_amd import gp as G
    from gogp_amd import priors
    noise = kernel.ScaledNoise(0.01)
    o = OracleGP(1, HYPER_SIMIL, noise)
    want, _ = _run(o, golden_dir, model=G.Model(o, priors.HyperPriors()), data="hyperpriors.csv", ntheta=6,
                   SEED=21, ITERS=4)
    g = G.GP(1, HYPER_SIMIL, noise)
    got, _ = _run(g, golden_dir, model=G.Model(g, priors.HyperPriors()), data="hyperpriors.csv", ntheta=6,
                  SEED=21, ITERS=4)
    assert len(got) == len(want) == 44
    for a, b in zip(got, want):
        assert abs(a[4] - b[4]) <= 2e-6 * max(1.0, abs(b[4])), (a, b)   # LML + log prior at the start
        assert abs(a[5] - b[5]) <= 1e-4 * max(1.0, abs(b[5])), (a, b)   # ... after 4 L-BFGS iterations
        assert abs(a[2] - b[2]) <= 1e-3 * max(1.0, abs(b[2])), (a, b)   # forecast mean
    from cases import ANYNOISE
    _, D, simil, anoise, _, _ = ANYNOISE
    o = OracleGP(D, simil, anoise)
    want, _ = _run(o, golden_dir, model=priors.AnyNoiseModel(G.Model(o, priors.AnyNoisePriors())), ntheta=3,
                   SEED=22, OPTINP=True, ITERS=3)
    g = G.GP(D, simil, anoise)
    got, _ = _run(g, golden_dir, model=priors.AnyNoiseModel(G.Model(g, priors.AnyNoisePriors())), ntheta=3,
                  SEED=22, OPTINP=True, ITERS=3)
    assert len(got) == len(want) == 20
    for a, b in zip(got, want):
        assert abs(a[4] - b[4]) <= 2e-6 * max(1.0, abs(b[4])), (a, b)
        assert abs(a[5] - b[5]) <= 1e-3 * max(1.0, abs(b[5])), (a, b)
