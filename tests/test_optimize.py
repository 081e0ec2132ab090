"""The L-BFGS hyperparameter loop and gp.Model (gp/model.go)."""
import numpy as np
import pytest

from gogp_amd import kernel, optimize


class Quadratic:
    """Elemental model with a known maximum (CPU test double for Observe/Gradient)."""

    def __init__(self, A, b):
        self.A, self.b = A, b
        self.calls = 0

    def Observe(self, x):
        self.calls += 1
        self._x = np.asarray(x, dtype=float)
        return float(-0.5 * self._x @ self.A @ self._x + self.b @ self._x)

    def Gradient(self):
        return -self.A @ self._x + self.b


def test_lbfgs_finds_quadratic_maximum():
    rng = np.random.default_rng(0)
    M = rng.normal(size=(6, 6))
    A = M @ M.T + 6 * np.eye(6)
    b = rng.normal(size=6)
    m = Quadratic(A, b)
    r = optimize.lbfgs(m, np.zeros(6), gradient_threshold=1e-9)
    assert r.converged
    np.testing.assert_allclose(r.x, np.linalg.solve(A, b), rtol=1e-7, atol=1e-9)
    assert r.evaluations == m.calls
    assert all(b2 >= a2 - 1e-12 for a2, b2 in zip(r.history, r.history[1:]))  # monotone ascent


def test_func_grad_negates_like_infer_funcgrad():
    m = Quadratic(np.eye(2), np.array([1.0, -2.0]))
    f, g = optimize.func_grad(m)
    x = np.array([0.3, 0.4])
    assert f(x) == -m.Observe(x)
    np.testing.assert_allclose(g(x), -(m.Gradient()))


def test_normal_log_priors_gradient():
    p = optimize.NormalLogPriors([0.0, 1.0], [1.0, 0.5])
    x = np.array([0.2, 0.7, 9.0])
    v = p.Observe(x)
    gr = p.Gradient()
    for i in range(2):
        h = 1e-6
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        assert abs((p.Observe(xp) - p.Observe(xm)) / (2 * h) - gr[i]) < 1e-8
    assert np.isfinite(v)


@pytest.mark.gpu
def test_lbfgs_on_gpu_matches_scipy_on_oracle():
    """Same start, same objective: the GPU path driven by our L-BFGS reaches the LML
    maximum that scipy's L-BFGS-B finds on the CPU oracle."""
    import scipy.optimize as so
    from gogp_amd import gp as G
    from gogp_amd import synth
    from oracle.oracle import FastOracle
    n, D = 600, 2
    X, y = synth.make_inputs(n, D, 77)
    simil, noise = kernel.Scaled(kernel.Normal), kernel.UniformNoise
    g = G.GP(D, simil, noise, X=X, Y=y)
    x0 = np.log([1.0, 0.5, 0.5])
    r = optimize.lbfgs(g, x0, gradient_threshold=1e-5, major_iterations=200)
    o = FastOracle(D, simil, noise)
    o.set_data(X, y)

    def fg(x):
        v = o.Observe(x)
        return -v, -o.Gradient()

    ref = so.minimize(fg, x0, jac=True, method="L-BFGS-B", options={"gtol": 1e-8, "maxiter": 500})
    assert abs(r.lml - (-ref.fun)) <= 1e-6 * abs(ref.fun), (r.lml, -ref.fun)
    np.testing.assert_allclose(r.x, ref.x, rtol=1e-3, atol=1e-3)
    assert np.abs(r.grad).max() < 1e-3


@pytest.mark.gpu
def test_gp_model_with_priors():
    """gp.Model (gp/model.go:9-28): Observe sums GP and prior log-densities, Gradient
    sums their gradients."""
    from gogp_amd import gp as G
    from gogp_amd import synth
    n, D = 200, 1
    X, y = synth.make_inputs(n, D, 5)
    g = G.GP(D, kernel.Scaled(kernel.Matern52), kernel.UniformNoise, X=X, Y=y)
    pri = optimize.NormalLogPriors([0.0, -1.0, -2.0], [1.0, 1.0, 1.0])
    m = G.Model(g, pri)
    x = np.array([0.1, -0.8, -1.5])
    ll = m.Observe(x)
    gr = m.Gradient()
    assert abs(ll - (g.Observe(x) + pri.Observe(x))) < 1e-9
    for i in range(3):
        h = 1e-5
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        fd = (m.Observe(xp) - m.Observe(xm)) / (2 * h)
        assert abs(fd - gr[i]) <= 1e-5 * max(1.0, abs(fd))
    r = optimize.lbfgs(m, x, gradient_threshold=1e-5, major_iterations=100)
    assert r.lml >= ll


class _BatchedQuadratic(Quadratic):
    """Quadratic with the candidates call of gogp_amd.gp.GP (CPU double)."""

    def observe_gradient_candidates(self, xs):
        xs = np.asarray(xs, dtype=float)
        self.batches = getattr(self, "batches", 0) + 1
        lmls = np.array([float(-0.5 * x @ self.A @ x + self.b @ x) for x in xs])
        grads = np.array([-self.A @ x + self.b for x in xs])
        return lmls, grads, np.zeros(len(xs), dtype=int)


def test_lbfgs_batched_line_search_takes_the_same_path():
    """line_search_candidates = k: the trial steps of a backtracking search evaluated k at a time
    accept exactly the points the one-at-a-time search accepts."""
    rng = np.random.default_rng(1)
    M = rng.normal(size=(5, 5))
    A = M @ M.T + 0.3 * np.eye(5)  # badly scaled: the line search backtracks
    b = rng.normal(size=5) * 5
    r1 = optimize.lbfgs(Quadratic(A, b), np.zeros(5), gradient_threshold=1e-7)
    m4 = _BatchedQuadratic(A, b)
    r4 = optimize.lbfgs(m4, np.zeros(5), gradient_threshold=1e-7, line_search_candidates=4)
    assert r4.converged and r1.converged and r4.iterations == r1.iterations
    np.testing.assert_allclose(r4.x, r1.x, rtol=0, atol=1e-13)
    np.testing.assert_allclose(r4.history, r1.history, rtol=1e-13)
    assert m4.batches <= r1.evaluations
    with pytest.raises(ValueError):
        optimize.lbfgs(Quadratic(A, b), np.zeros(5), line_search_candidates=4)


@pytest.mark.gpu
def test_lbfgs_batched_line_search_on_gpu_is_bit_identical():
    """On the GPU the candidates of one launch sequence are bit for bit the single evaluations,
    so the batched line search returns the identical optimum -- for a GP and for a gp.Model with
    priors -- and leaves the GP at it."""
    from gogp_amd import gp as G
    from gogp_amd import synth
    n, D = 700, 2
    X, y = synth.make_inputs(n, D, 78)
    simil, noise = kernel.Scaled(kernel.Matern52), kernel.UniformNoise
    x0 = np.log([2.0, 0.2, 0.7])
    g1 = G.GP(D, simil, noise, X=X, Y=y)
    r1 = optimize.lbfgs(g1, x0, gradient_threshold=1e-5, major_iterations=60)
    g4 = G.GP(D, simil, noise, X=X, Y=y)
    r4 = optimize.lbfgs(g4, x0, gradient_threshold=1e-5, major_iterations=60, line_search_candidates=4)
    np.testing.assert_array_equal(r4.x, r1.x)
    assert r4.lml == r1.lml and r4.iterations == r1.iterations
    assert g4.LML() == r4.lml  # the GP holds the factorisation of the returned point
    pri = optimize.NormalLogPriors([0.0, -1.0, -2.0], [1.0, 1.0, 1.0])
    m1, m4 = G.Model(g1, pri), G.Model(g4, optimize.NormalLogPriors([0.0, -1.0, -2.0], [1.0, 1.0, 1.0]))
    q1 = optimize.lbfgs(m1, x0, gradient_threshold=1e-5, major_iterations=60)
    q4 = optimize.lbfgs(m4, x0, gradient_threshold=1e-5, major_iterations=60, line_search_candidates=3)
    np.testing.assert_allclose(q4.x, q1.x, rtol=0, atol=1e-12)
    assert abs(q4.lml - q1.lml) <= 1e-12 * abs(q1.lml)
    g1.close()
    g4.close()


def test_lbfgs_multistart_runs_take_the_paths_of_single_runs():
    """k restarts in lock-step (one batched evaluation per round) = k separate lbfgs runs."""
    rng = np.random.default_rng(2)
    M = rng.normal(size=(4, 4))
    A = M @ M.T + 0.5 * np.eye(4)
    b = rng.normal(size=4) * 3
    starts = rng.normal(size=(3, 4))
    singles = [optimize.lbfgs(Quadratic(A, b), x0, gradient_threshold=1e-7) for x0 in starts]
    m = _BatchedQuadratic(A, b)
    multi = optimize.lbfgs_multistart(m, starts, gradient_threshold=1e-7)
    for a, c in zip(multi, singles):
        assert a.converged and a.iterations == c.iterations and a.evaluations == c.evaluations
        np.testing.assert_allclose(a.x, c.x, rtol=0, atol=1e-13)
        np.testing.assert_allclose(a.history, c.history, rtol=1e-13)
    assert m.batches <= max(c.evaluations for c in singles)  # rounds, not evaluations


@pytest.mark.gpu
def test_lbfgs_multistart_on_gpu():
    """Three restarts on the HIP path: each run is bit for bit the single run from its start, the GP
    ends at the best optimum."""
    from gogp_amd import gp as G
    from gogp_amd import synth
    n, D = 500, 2
    X, y = synth.make_inputs(n, D, 79)
    simil, noise = kernel.Scaled(kernel.Matern52), kernel.UniformNoise
    starts = np.log([[2.0, 0.2, 0.7], [0.5, 1.0, 0.3], [1.0, 0.5, 0.5]])
    g = G.GP(D, simil, noise, X=X, Y=y)
    singles = [optimize.lbfgs(g, x0, gradient_threshold=1e-5, major_iterations=40) for x0 in starts]
    multi = optimize.lbfgs_multistart(g, starts, gradient_threshold=1e-5, major_iterations=40)
    for a, c in zip(multi, singles):
        np.testing.assert_array_equal(a.x, c.x)
        assert a.lml == c.lml and a.iterations == c.iterations
    best = max(multi, key=lambda r: r.lml)
    assert g.LML() == best.lml
    g.close()
