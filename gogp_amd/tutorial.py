"""The tutorial's forecast harness over the HIP path ("next" row 2 of SURVEY.md 8f).

Mirror of tutorial/tutorial.go: ``Evaluate`` (:56-230) -- load a CSV, standardise the
outputs, and for every time point fit the hyperparameters on the points before it and
forecast it one step out of sample -- and ``load`` (:234-272).  Same knobs (module
variables, the reference's package variables :21-33), same output columns, same number
formatting (``%f``), so the case studies of the tutorial run unchanged on a
gogp_amd.gp.GP / gp.Model.

Differences, all forced by what is absent here: the optimiser is gogp_amd.optimize.lbfgs /
Adam instead of gonum's ``optimize.Minimize`` / infergo's ``infer.Adam`` (sources not in the
container: iterate-by-iterate parity unpinned), and the random jitter of the starting
point (:119-121, seeded from the wall clock at :36) takes a ``SEED`` so that runs can be
repeated; SEED = None seeds from the clock as the reference does.
"""
from __future__ import annotations

import csv
import math
import sys
import time
from typing import Optional

import numpy as np

from . import optimize

# package variables of tutorial/tutorial.go:21-33
OPTINP = False
MINOPT = 0
ALG = "lbfgs"
PARALLEL = False
ITERS = 1000       # major iterations
MINITERS = 10      # minimum iterations to accept in lbfgs
THRESHOLD = 1e-6   # gradient threshold
RATE = 0.01        # learning rate (for Adam)
NTASKS = 0
NONORMALIZE = False
OUTOFSAMPLE = False
SEED: Optional[int] = None  # not in the reference: fixes the starting-point jitter


def _f(v: float) -> str:
    """Go's fmt %f."""
    v = float(v)
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "+Inf" if v > 0 else "-Inf"
    return "%f" % v


def load(rdr):
    """tutorial/tutorial.go:234-272: every record is D inputs followed by one output.
    Returns (X as an (n, D) array, y); raises ValueError on a field that is not a number
    (the reference returns the strconv error)."""
    X, y = [], []
    for record in csv.reader(rdr):
        if not record:
            continue
        row = [float(f) for f in record]  # ValueError = the reference's data error
        X.append(row[:-1])
        y.append(row[-1])
    ndim = len(X[0]) if X else 0
    return np.array(X, dtype=float).reshape(len(X), ndim), np.array(y, dtype=float)


def Evaluate(gp, m, theta, rdr, wtr, log=sys.stderr) -> None:
    """tutorial/tutorial.go:56-230.  ``gp``: a GP (NDim, X, Y, Produce); ``m``: the model that is
    optimised (the GP itself or a gp.Model around it); ``theta``: initial LOG hyperparameters;
    ``rdr`` / ``wtr``: text streams of the CSV data and of the forecasts."""
    gp.Parallel = bool(PARALLEL)
    rng = np.random.default_rng(time.time_ns() if SEED is None else SEED)
    theta = np.asarray(theta, dtype=float)

    print("loading...", end="", file=log)
    X, Y = load(rdr)
    print("done", file=log)

    # Normalize Y (gonum stat.MeanStdDev: the unbiased, n-1, standard deviation)
    if NONORMALIZE:
        meany, stdy = 0.0, 1.0
    else:
        meany = float(Y.mean()) if len(Y) else 0.0
        stdy = float(Y.std(ddof=1)) if len(Y) > 1 else float("nan")
        Y = (Y - meany) / stdy

    print("Forecasting...", file=log)
    x = theta.copy()
    for end in range(len(X)):
        Xi, Yi = X[:end], Y[:end]
        if OPTINP:
            # inputs and outputs ride in the parameter vector of Observe (:100-110)
            x = np.concatenate([theta, Xi.reshape(-1), Yi])
        else:
            x = theta.copy()
            gp.X, gp.Y = Xi, Yi
        # Randomize the initial values of hyperparameters (:119-121)
        x[:len(theta)] += 0.1 * rng.standard_normal(len(theta))

        lml0 = m.Observe(x)  # Initial log likelihood

        if len(gp.X) > MINOPT:
            if ALG == "lbfgs":
                try:
                    # optimize.Settings.Concurrent = NTASKS (tutorial.go:141): evaluate that many trial
                    # points per round -- here the line search's next NTASKS steps in one launch
                    # sequence (hyperparameters-only form on a GP / gp.Model with the candidates call)
                    conc = NTASKS if (NTASKS > 1 and not OPTINP and
                                      hasattr(getattr(m, "GP", m), "observe_gradient_candidates")) else 1
                    result = optimize.lbfgs(m, x, major_iterations=ITERS, gradient_threshold=THRESHOLD,
                                            line_search_candidates=conc)
                    if not result.converged and result.iterations <= MINITERS:
                        print("%d: stuck after %d iterations" % (end, result.iterations), file=log)
                    x = result.x
                except ValueError as e:  # infeasible start
                    print("%d: stuck after 0 iterations: %s" % (end, e), file=log)
            elif ALG == "adam":
                opt = optimize.Adam(Rate=RATE)
                for _ in range(ITERS):
                    _, grad = opt.Step(m, x)
                    if not (np.abs(grad) >= THRESHOLD).any():
                        break
            else:
                raise ValueError("ALG must be lbfgs or adam")

        lml = m.Observe(x)  # Final log likelihood

        Z = X[end:end + 1]
        try:
            mu, sigma = gp.Produce(Z)
        except Exception as e:  # the reference prints and carries on (:179-181)
            print("Failed to forecast: %s" % e, file=log)
            mu, sigma = [float("nan")], [float("nan")]

        fields = [_f(v) for v in Z[0]]
        fields += [_f(Y[end] * stdy + meany), _f(mu[0] * stdy + meany), _f(sigma[0] * stdy),
                   _f(lml0), _f(lml)]
        fields += [_f(math.exp(v)) for v in x[:len(theta)]]
        wtr.write(",".join(fields) + "\n")

    if OUTOFSAMPLE and len(X):
        Z = (X + X[-1])[1:]  # :200-208
        try:
            mu, sigma = gp.Produce(Z)
        except Exception as e:
            print("Failed to forecast: %s" % e, file=log)
            mu = sigma = np.full(len(Z), float("nan"))
        for i in range(len(Z)):
            fields = [_f(v) for v in Z[i]] + ["nan", _f(mu[i] * stdy + meany), _f(sigma[i] * stdy)]
            wtr.write(",".join(fields) + "\n")

    print("done", file=log)
