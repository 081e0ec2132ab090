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
             ("OPTINP", "MINOPT", "ALG", "ITERS", "THRESHOLD", "RATE", "NONORMALIZE", "OUTOFSAMPLE", "SEED")}
    yield tutorial
    for k, v in saved.items():
        setattr(tutorial, k, v)


DEFAULTS = dict(OPTINP=False, MINOPT=0, ALG="lbfgs", ITERS=1000, THRESHOLD=1e-6, RATE=0.01,
                NONORMALIZE=False, OUTOFSAMPLE=False, SEED=None)


def _run(gp, golden_dir, **kn):
    for k, v in dict(DEFAULTS, **kn).items():
        setattr(tutorial, k, v)
    out = io.StringIO()
    with open(os.path.join(golden_dir, "barebones.csv")) as f:
        tutorial.Evaluate(gp, gp, np.zeros(3), f, out, log=io.StringIO())
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
    # full form (inputs and outputs in x), a few steps
    want, _ = _run(OracleGP(1, SIMIL, NOISE), golden_dir, SEED=13, OPTINP=True, ITERS=2)
    got, _ = _run(G.GP(1, SIMIL, NOISE), golden_dir, SEED=13, OPTINP=True, ITERS=2)
    for g, w in zip(got, want):
        assert abs(g[5] - w[5]) <= 1e-5 * max(1.0, abs(w[5])), (g, w)
