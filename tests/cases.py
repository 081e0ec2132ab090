"""Kernel families exercised by the parity tests and by the committed oracle vectors
(tests/golden/make_oracle_vectors.py): (name, NDim, Simil, Noise, theta_simil, theta_noise).

The primitives are the reference's (kernel/kernel.go:23-26,44-47,70-73,89-92,
kernel/noise.go:27-30,47-49); the compositions mirror its tutorials
(tutorial/barebones/kernel/kernel.go:14-31, tutorial/hyperpriors/kernel/kernel.go:23-24).
"""
from gogp_amd import kernel

CASES = [
    ("normal1d", 1, kernel.Normal, kernel.ConstantNoise(0.1), [0.3], []),
    ("scaled_rbf", 4, kernel.Scaled(kernel.Normal), kernel.UniformNoise, [1.0, 0.8], [0.1]),
    ("ard_rbf", 5, kernel.Scaled(kernel.ARD(kernel.Normal, 5)), kernel.UniformNoise,
     [1.2, 0.9, 1.0, 1.1, 1.2, 1.3], [0.2]),
    ("matern32", 2, kernel.Scaled(kernel.Matern32), kernel.ScaledNoise(0.01), [1.0, 0.7], [1.5]),
    ("matern52_ref", 3, kernel.Scaled(kernel.Matern52), kernel.UniformNoise, [0.9, 1.1], [0.15]),
    ("matern52_textbook", 3, kernel.Scaled(kernel.Matern52Textbook), kernel.UniformNoise,
     [0.9, 1.1], [0.15]),
    ("periodic", 1, kernel.Scaled(kernel.Periodic), kernel.UniformNoise, [1.0, 0.8, 0.45], [0.2]),
    ("hyperpriors", 1,
     kernel.Sum([kernel.Scaled(kernel.Matern52), kernel.Scaled(kernel.PeriodScaled(kernel.Periodic, 10.0))],
                order=[0, 2, 1, 3, 4]), kernel.ScaledNoise(0.01), [1.0, 0.5, 0.6, 1.3, 0.05], [2.0]),
    ("default_noise", 2, kernel.Scaled(kernel.Matern32), None, [1.0, 0.3], []),
]

#: tutorial/anynoise/kernel/kernel.go:12-35: c*Matern52 with a constant 1e-5 noise that still
#: owns one parameter (used by the priors only).  Kept out of CASES: the committed
#: oracle_vectors.json enumerates CASES.
ANYNOISE = ("anynoise", 1, kernel.Scaled(kernel.Matern52), kernel.ConstantNoiseParam(1e-5 ** 0.5),
            [1.1, 0.6], [0.3])
