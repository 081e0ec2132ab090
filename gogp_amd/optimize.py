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
    x = np.array(x0, dtype=float)
    evals = 0
    kls = int(line_search_candidates)
    gp_b = getattr(m, "GP", m)
    priors_b = getattr(m, "Priors", None) if hasattr(m, "GP") else None
    if kls > 1 and not hasattr(gp_b, "observe_gradient_candidates"):
        raise ValueError("line_search_candidates needs a GP with observe_gradient_candidates")

    def batch_value_and_grad(xs):
        """(-LML - log prior, its gradient) per row of xs; inf / None where K is not usable."""
        nonlocal evals
        evals += len(xs)
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

    def value_and_grad(xx):
        nonlocal evals
        evals += 1
        try:
            v = m.Observe(xx)
        except Exception as e:
            # Only "K is not positive definite at this trial point" (gp/gp.go:228-230) is an
            # infeasible point the line search may back off from; anything else (a HIP error,
            # a bad argument) is a real failure and must surface.
            if not _is_not_positive_definite(e):
                raise
            return np.inf, None
        if not np.isfinite(v):
            return np.inf, None
        return -v, None

    f, _ = value_and_grad(x)
    if not np.isfinite(f):
        raise ValueError("initial point is not feasible")
    g = -np.asarray(m.Gradient(), dtype=float)
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
        if kls > 1:
            trials = 0
            while trials < 30 and not accepted:
                kk = min(kls, 30 - trials)
                steps_c = step * 0.5 ** np.arange(kk)
                res = batch_value_and_grad(x[None, :] + steps_c[:, None] * d[None, :])
                for c in range(kk):
                    fc, gc = res[c]
                    if np.isfinite(fc) and fc <= f + 1e-4 * steps_c[c] * slope:
                        accepted, step, fn, gn = True, steps_c[c], fc, gc
                        xn = x + step * d
                        break
                trials += kk
                if not accepted:
                    step = steps_c[-1] * 0.5
        else:
            for _ in range(30):
                xn = x + step * d
                fn, _ = value_and_grad(xn)
                if np.isfinite(fn) and fn <= f + 1e-4 * step * slope:
                    accepted = True
                    break
                step *= 0.5
        if not accepted:
            break
        if gn is None:
            gn = -np.asarray(m.Gradient(), dtype=float)
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
    if kls > 1:  # the candidates never touched the GP's own state: leave it at the returned point
        m.Observe(x)
    return Result(x=x, lml=-f, grad=-g, iterations=it, evaluations=evals, converged=converged,
                  history=hist)


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
