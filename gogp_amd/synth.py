"""Synthetic inputs of the benchmark configurations.

Counter-based generator: splitmix64 of (seed, index) -> U[0,1) double, so any
rank / any language reproduces the same bits without shared state.

    X[i][d] ~ U(0,1)
    y_i     = sum_d sin(2 pi x_id)/sqrt(D) + 0.1 * N(0,1)   (Box-Muller, same stream)
    y       standardised (mirrors tutorial/tutorial.go:78-86)
    theta   c = 1, l = sqrt(D/6), sigma = 0.1  (K = c*k + sigma^2 I)
"""
from __future__ import annotations

import math

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, start: int, count: int) -> np.ndarray:
    """U[0,1) doubles number start .. start+count-1 of stream `seed`."""
    idx = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = splitmix64(np.full(1, seed, dtype=np.uint64))[0]
        bits = splitmix64(idx * np.uint64(0x2545F4914F6CDD1D) + key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_inputs(n: int, ndim: int, seed: int):
    """(X, y) of the benchmark workload; y standardised."""
    X = uniform01(seed, 0, n * ndim).reshape(n, ndim)
    u1 = uniform01(seed + 1, 0, n)
    u2 = uniform01(seed + 2, 0, n)
    gauss = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)
    y = np.sin(2.0 * math.pi * X).sum(axis=1) / math.sqrt(ndim) + 0.1 * gauss
    y = (y - y.mean()) / y.std()
    return X, y


def make_test_points(m: int, ndim: int, seed: int) -> np.ndarray:
    return uniform01(seed + 7, 0, m * ndim).reshape(m, ndim)


def theta0(ndim: int):
    """[c, l, sigma] natural scale for Scaled(RBF) + UniformNoise."""
    return np.array([1.0, math.sqrt(ndim / 6.0), 0.1])


def log_theta_cycle(ndim: int, step: int, rank: int = 0) -> np.ndarray:
    """log theta of evaluation `step`: a fixed +-1 % cycle around theta0 so that
    no two consecutive evaluations share hyperparameters (nothing can be memoised);
    ranks use shifted phases (independent candidates)."""
    th = theta0(ndim)
    k = step + 3 * rank
    f = np.array([1.0 + 0.01 * ((k % 5) - 2) / 2.0,
                  1.0 + 0.01 * (((k + 1) % 3) - 1),
                  1.0 + 0.01 * (((k + 2) % 4) - 1.5) / 1.5])
    return np.log(th * f)
