"""The five workload configurations of BASELINE.json (``configs[0..4]``), as data:
kernel family, N, D, arithmetic type, parallelism, synthetic inputs and the
hyperparameters every measurement is quoted at.

    1  tutorial/barebones: 1-D Normal kernel, N=64 (x_i = i*pi/10, y = sin x + noise: the
       recipe of the reference's 20-row tutorial/data/barebones.csv extended to 64 rows)
    2  RBF + homoscedastic noise, N=4096  D=4   fp64, one GPU
    3  RBF + white noise,         N=16384 D=8   fp64, one GPU      (the headline metric)
    4  Matern-5/2,                N=32768 D=16  fp64, 2-D block-cyclic over the GPUs
    5  ARD-RBF,                   N=65536 D=32  fp32, 2-D block-cyclic, L-BFGS loop

The reference ships 1-D primitives only (kernel/kernel.go:15-17); the D-dimensional
forms are r^2 = sum_d ((xa_d - xb_d)/l_d)^2 substituted into the reference formulas.
Matern-5/2 uses the reference's coefficient (Go's integer 5/3 == 1, kernel/kernel.go:91).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import kernel, synth

SEED0 = 20251114  # seed = 20251114 + config index (0-based)


@dataclass
class Workload:
    config: int
    name: str
    N: int
    D: int
    simil: object
    noise: object
    theta: np.ndarray          # natural scale [ThetaSimil | ThetaNoise]
    dtype: str                 # arithmetic type the path computes in
    sharded: bool              # BASELINE quotes it as ONE evaluation over all GPUs
    kernel_text: str
    seed: int

    @property
    def P(self) -> int:
        return len(self.theta)

    def inputs(self, n: Optional[int] = None):
        n = self.N if n is None else n
        if self.config == 1:
            # tutorial/data/barebones.csv recipe: x = i*pi/10, y = sin(x) + noise, standardised
            # as tutorial/tutorial.go:78-86 does
            x = np.arange(n, dtype=float) * (math.pi / 10.0)
            u1 = synth.uniform01(self.seed + 1, 0, n)
            u2 = synth.uniform01(self.seed + 2, 0, n)
            y = np.sin(x) + 0.1 * np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)
            y = (y - y.mean()) / y.std()
            return x.reshape(n, 1), y
        X, y = synth.make_inputs(self.N, self.D, self.seed)
        return X[:n], y[:n]  # prefix property of the counter-based stream

    def test_points(self, m: int):
        if self.config == 1:
            return (np.arange(m, dtype=float) + 0.5).reshape(m, 1) * (math.pi / 10.0)
        return synth.make_test_points(m, self.D, self.seed + 1)

    def log_theta(self, step: int, rank: int = 0) -> np.ndarray:
        """log theta of evaluation `step`: a fixed +-1 % cycle around `theta`, so that no
        two consecutive evaluations share hyperparameters (nothing can be memoised)."""
        if self.config == 3:
            return synth.log_theta_cycle(self.D, step, rank)
        k = step + 3 * rank
        p = np.arange(self.P)
        f = 1.0 + 0.01 * (((k + 2 * p) % 5) - 2) / 2.0
        return np.log(self.theta * f)


def workload(config: int, nobs: Optional[int] = None, ndim: Optional[int] = None) -> Workload:
    if config == 1:
        N, D = nobs or 64, 1
        return Workload(1, "BASELINE configs[0]: tutorial/barebones recipe, 1-D Normal kernel, N=%d" % N,
                        N, D, kernel.Normal, kernel.ScaledNoise(0.01), np.array([1.0, 1.0]), "f64",
                        False, "RBF(l) + 0.01 sigma^2 I", SEED0 + 0)
    if config == 2:
        N, D = nobs or 4096, ndim or 4
        return Workload(2, "BASELINE configs[1]: RBF + homoscedastic noise, N=%d D=%d fp64" % (N, D),
                        N, D, kernel.Scaled(kernel.Normal), kernel.UniformNoise,
                        np.array([1.0, math.sqrt(D / 6.0), 0.1]), "f64", False,
                        "c*RBF(l) + sigma^2 I", SEED0 + 1)
    if config == 3:
        N, D = nobs or 16384, ndim or 8
        return Workload(3, "BASELINE configs[2]: RBF + white noise, N=%d D=%d fp64" % (N, D),
                        N, D, kernel.Scaled(kernel.Normal), kernel.UniformNoise,
                        synth.theta0(D), "f64", False, "c*RBF(l) + sigma^2 I", SEED0 + 2)
    if config == 4:
        N, D = nobs or 32768, ndim or 16
        return Workload(4, "BASELINE configs[3]: Matern-5/2, N=%d D=%d fp64" % (N, D),
                        N, D, kernel.Scaled(kernel.Matern52), kernel.UniformNoise,
                        np.array([1.0, math.sqrt(D / 6.0), 0.1]), "f64", True,
                        "c*Matern52(l) [reference coefficient] + sigma^2 I", SEED0 + 3)
    if config == 5:
        N, D = nobs or 65536, ndim or 32
        ls = math.sqrt(D / 6.0) * (1.0 + np.arange(D) / (2.0 * D))
        return Workload(5, "BASELINE configs[4]: ARD-RBF, N=%d D=%d" % (N, D),
                        N, D, kernel.Scaled(kernel.ARD(kernel.Normal, D)), kernel.UniformNoise,
                        np.concatenate([[1.0], ls, [0.1]]), "f32", True,
                        "c*ARD-RBF(l_1..l_D) + sigma^2 I", SEED0 + 4)
    raise ValueError("config must be 1..5")
