"""Hyperparameter optimisation loop over Observe/Gradient ("next" row 1 of
SURVEY.md section 8f).

The reference drives ``m.Observe`` / ``model.Gradient(m)`` through
``infer.FuncGrad`` and gonum's ``optimize.Minimize`` with the default method for
problems with a gradient (L-BFGS), ``MajorIterations = 1000`` and
``GradientThreshold = 1e-6`` (tutorial/tutorial.go:128-155).  gonum's source is
not available here, so iterate-by-iterate parity is unpinned; this module is a
plain L-BFGS (two-loop recursion, backtracking/Armijo + curvature-safeguarded
line search) on the NEGATIVE log marginal likelihood with the same stopping
rule, so that final LML / theta can be compared from identical starts.

Every function value needs one Observe; the gradient of the accepted point comes
from the same evaluation (the fused sweep prepares it during Observe), so an
iteration with an accepted first trial costs exactly one Observe + Gradient.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np


@dataclass
class Result:
    x: np.ndarray
    lml: float
    grad: np.ndarray
    iterations: int
    evaluations: int
    converged: bool
    history: List[float] = field(default_factory=list)


def func_grad(m):
    """infer.FuncGrad(m) (tutorial/tutorial.go:131): closures returning the
    NEGATED log-likelihood and gradient of an elemental model."""

    def f(x):
        return -m.Observe(x)

    def g(x):
        m.Observe(x)
        return -np.asarray(m.Gradient())

    return f, g


def _is_not_positive_definite(e: Exception) -> bool:
    if isinstance(e, (np.linalg.LinAlgError, ArithmeticError)):
        return True
    if getattr(e, "code", None) in (2, 6):  # GOGP_ENOTPD / GOGP_ECOND: K not usable at this point
        return True
    return type(e).__name__ in ("FactorizeError", "ConditionError", "NotPositiveDefinite")


def _lbfgs_steps(x0, major_iterations, gradient_threshold, history_size, max_step_log, kls, callback):
    """The L-BFGS iteration as a generator: it yields requests and is sent their answers, so that
    one algorithm serves the sequential driver, the batched line search and the lock-step
    multi-start driver.  Requests:
      ("eval", X[k x P], want_grad)  ->  list of k (f, g) with f = -objective (inf where K is not
                                         usable) and g = -gradient or None (when not wanted and
                                         not available for free)
      ("grad", c)                     ->  g of row c of the LAST "eval" (only asked when that "eval"
                                         returned None for it)
    Returns the Result through StopIteration."""
    x = np.array(x0, dtype=float)
    evals = 1
    (f, g), = yield ("eval", x[None, :], True)
    if not np.isfinite(f):
        raise ValueError("initial point is not feasible")
    S: List[np.ndarray] = []
    Yv: List[np.ndarray] = []
    hist = [-f]
    converged = False
    it = 0
    for it in range(1, major_iterations + 1):
        if np.abs(g).max() <= gradient_threshold:
            converged = True
            break
        # two-loop recursion
        q = g.copy()
        al = []
        for s, yv in zip(reversed(S), reversed(Yv)):
            rho = 1.0 / float(yv @ s)
            a = rho * float(s @ q)
            al.append((a, rho, s, yv))
            q -= a * yv
        if S:
            gamma = float(S[-1] @ Yv[-1]) / float(Yv[-1] @ Yv[-1])
            q *= gamma
        for a, rho, s, yv in reversed(al):
            b = rho * float(yv @ q)
            q += (a - b) * s
        d = -q
        if float(d @ g) >= 0:  # not a descent direction: reset
            d = -g
            S.clear()
            Yv.clear()
        # cap the step in log-theta space (keeps K positive definite in practice)
        step = 1.0
        dmax = np.abs(d).max()
        if dmax * step > max_step_log:
            step = max_step_log / dmax
        if not S:
            step = min(step, 1.0 / max(1.0, np.abs(g).max()))
        slope = float(d @ g)
        accepted = False
        gn = None
        trials = 0
        while trials < 30 and not accepted:
            kk = min(kls, 30 - trials)
            steps_c = step * 0.5 ** np.arange(kk)
            res = yield ("eval", x[None, :] + steps_c[:, None] * d[None, :], False)
            evals += kk
            for c in range(kk):
                fc, gc = res[c]
                if np.isfinite(fc) and fc <= f + 1e-4 * steps_c[c] * slope:
                    accepted, step, fn, gn = True, steps_c[c], fc, gc
                    xn = x + step * d
                    if gn is None:
                        gn = yield ("grad", c)
                    break
            trials += kk
            if not accepted:
                step = steps_c[-1] * 0.5
        if not accepted:
            break
        s_vec, y_vec = xn - x, gn - g
        if float(s_vec @ y_vec) > 1e-12 * float(np.linalg.norm(s_vec) * np.linalg.norm(y_vec)):
            S.append(s_vec)
            Yv.append(y_vec)
            if len(S) > history_size:
                S.pop(0)
                Yv.pop(0)
        x, f, g = xn, fn, gn
        hist.append(-f)
        if callback:
            callback(it, x, -f)
    return Result(x=x, lml=-f, grad=-g, iterations=it, evaluations=evals, converged=converged,
                  history=hist)


def _batch_evaluator(m):
    """xs -> list of (-LML - log prior, its gradient) through GP.observe_gradient_candidates (one
    launch sequence for all rows); (inf, None) where K is not usable."""
    gp_b = getattr(m, "GP", m)
    priors_b = getattr(m, "Priors", None) if hasattr(m, "GP") else None
    if not hasattr(gp_b, "observe_gradient_candidates"):
        raise ValueError("needs a GP with observe_gradient_candidates")

    def evaluate(xs):
        lmls, grads, status = gp_b.observe_gradient_candidates(xs)
        out = []
        for c in range(len(xs)):
            if status[c] != 0 or not np.isfinite(lmls[c]):
                out.append((np.inf, None))
                continue
            v, gr = float(lmls[c]), np.array(grads[c], dtype=float)
            if priors_b is not None:  # gp/model.go:17-28
                v += priors_b.Observe(xs[c])
                pg = np.asarray(priors_b.Gradient(), dtype=float)
                gr[:len(pg)] += pg
            out.append((-v, -gr) if np.isfinite(v) else (np.inf, None))
        return out

    return evaluate


def lbfgs(m, x0, major_iterations: int = 1000, gradient_threshold: float = 1e-6,
          history_size: int = 10, max_step_log: float = 2.0,
          callback: Optional[Callable[[int, np.ndarray, float], None]] = None,
          line_search_candidates: int = 1) -> Result:
    """Maximise m.Observe(x) (LML) over x = log theta with L-BFGS.

    ``m`` is a gogp_amd.gp.GP, a gogp_amd.gp.Model, or anything with
    Observe(x)/Gradient().  Stops when ||grad||_inf <= gradient_threshold
    (gonum's GradientThreshold) or after major_iterations.

    ``line_search_candidates`` = k > 1: the backtracking line search evaluates its next k trial
    steps (step, step/2, ...) in ONE launch sequence (GP.observe_gradient_candidates) and takes the
    first that satisfies the Armijo condition -- the same accepted points, values and gradients
    as k = 1, bit for bit, in fewer and better-filled passes over the GPU (the reference's
    counterpart: optimize.Settings.Concurrent, tutorial/tutorial.go:30,141).  Hyperparameters-only
    form; the GP ends at the returned point."""
    kls = int(line_search_candidates)
    gen = _lbfgs_steps(x0, major_iterations, gradient_threshold, history_size, max_step_log, max(1, kls), callback)
    batch = _batch_evaluator(m) if kls > 1 else None

    def observe(xx):
        try:
            v = m.Observe(xx)
        except Exception as e:
            # Only "K is not positive definite at this trial point" (gp/gp.go:228-230) is an
            # infeasible point the line search may back off from; anything else (a HIP error,
            # a bad argument) is a real failure and must surface.
            if not _is_not_positive_definite(e):
                raise
            return np.inf
        return -v if np.isfinite(v) else np.inf

    try:
        req = next(gen)
        while True:
            if req[0] == "eval":
                xs, want_grad = req[1], req[2]
                if batch is not None:
                    ans = batch(xs)
                else:  # one point: Observe; its gradient only when asked for
                    fv = observe(xs[0])
                    gv = -np.asarray(m.Gradient(), dtype=float) if (want_grad and np.isfinite(fv)) else None
                    ans = [(fv, gv)]
                req = gen.send(ans)
            else:  # ("grad", c): the accepted trial is the point the model was last observed at
                req = gen.send(-np.asarray(m.Gradient(), dtype=float))
    except StopIteration as stop:
        res = stop.value
    if batch is not None:  # the candidates never touched the GP's own state: leave it at the returned point
        m.Observe(res.x)
    return res


def lbfgs_multistart(m, x0s, major_iterations: int = 1000, gradient_threshold: float = 1e-6,
                     history_size: int = 10, max_step_log: float = 2.0) -> List[Result]:
    """k L-BFGS runs from the rows of ``x0s`` in lock-step: every round evaluates the current trial
    point of every run that is still going in ONE launch sequence (GP.observe_gradient_candidates).
    Each run takes exactly the path ``lbfgs`` takes from its start alone (same values bit for bit);
    what changes is that the GPU sees k evaluations at a time.  The reference randomises the start
    of every fit (tutorial/tutorial.go:119-121) and lets gonum evaluate concurrently
    (optimize.Settings.Concurrent, :141); this is the restart loop that goes with it.  Returns the
    k Results (best: max(results, key=lambda r: r.lml)); the GP ends at the best point."""
    x0s = np.atleast_2d(np.asarray(x0s, dtype=float))
    batch = _batch_evaluator(m)
    gens = [_lbfgs_steps(x0, major_iterations, gradient_threshold, history_size, max_step_log, 1, None)
            for x0 in x0s]
    results: List[Optional[Result]] = [None] * len(gens)
    pending = {}
    for i, gen in enumerate(gens):
        pending[i] = next(gen)
    while pending:
        idx = sorted(pending)
        ans = batch(np.concatenate([pending[i][1] for i in idx], axis=0))
        for j, i in enumerate(idx):
            try:
                pending[i] = gens[i].send([ans[j]])
            except StopIteration as stop:
                results[i] = stop.value
                del pending[i]
            except ValueError:  # infeasible start: this run yields nothing
                del pending[i]
    done = [r for r in results if r is not None]
    if done:
        m.Observe(max(done, key=lambda r: r.lml).x)
    return results


class NormalLogPriors:
    """Independent Normal log-priors on the (log-scale) parameters, the shape of
    tutorial/hyperpriors/model/model.go: Observe sums log-densities, Gradient
    returns their derivatives (an elemental model for gp.Model.Priors)."""

    def __init__(self, mean, std):
        self.mean = np.asarray(mean, dtype=float)
        self.std = np.asarray(std, dtype=float)
        self._x = None

    def Observe(self, x) -> float:
        x = np.asarray(x, dtype=float)[:len(self.mean)]
        self._x = x
        z = (x - self.mean) / self.std
        return float((-0.5 * z * z - np.log(self.std) - 0.5 * np.log(2 * np.pi)).sum())

    def Gradient(self) -> np.ndarray:
        return -(self._x - self.mean) / (self.std ** 2)


class Adam:
    """infer.Adam{Rate: r} as the tutorial uses it (tutorial/tutorial.go:156-168): ``Step(m, x)``
    evaluates the model, moves x IN PLACE one Adam step UP the log-likelihood gradient and
    returns (log-likelihood, gradient) at the point it evaluated.  infergo v1.2.2 is not in
    the container; the update is the published Adam rule with its usual defaults
    (beta1 0.9, beta2 0.999, eps 1e-8) -- iterate-by-iterate parity is unpinned."""

    def __init__(self, Rate: float = 0.01, Beta1: float = 0.9, Beta2: float = 0.999, Eps: float = 1e-8):
        self.Rate, self.Beta1, self.Beta2, self.Eps = Rate, Beta1, Beta2, Eps
        self._m = None
        self._v = None
        self._t = 0

    def Step(self, m, x):
        ll = m.Observe(x)
        grad = np.asarray(m.Gradient(), dtype=float)
        if self._m is None:
            self._m = np.zeros_like(grad)
            self._v = np.zeros_like(grad)
        self._t += 1
        self._m = self.Beta1 * self._m + (1 - self.Beta1) * grad
        self._v = self.Beta2 * self._v + (1 - self.Beta2) * grad * grad
        mhat = self._m / (1 - self.Beta1 ** self._t)
        vhat = self._v / (1 - self.Beta2 ** self._t)
        x += self.Rate * mhat / (np.sqrt(vhat) + self.Eps)
        return ll, grad
