"""Accuracy of the 256-block factor/inverse kernel on matrices of several condition numbers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G
rng = np.random.default_rng(1)
for cond_shift in (256.0, 1.0, 1e-3, 1e-6):
    B = rng.normal(size=(256, 256))
    A = B @ B.T / 256 + cond_shift * np.eye(256)
    L, X, st, us = G.diag256_check(A)
    Lr = np.linalg.cholesky(A)
    eL = np.abs(L - Lr).max() / np.abs(Lr).max()
    eA = np.abs(L @ L.T - A).max() / np.abs(A).max()
    eX = np.abs(X @ Lr - np.eye(256)).max()
    print("shift %-8g cond %.2e: rel|L-Lref| %.2e  rel|LL^T-A| %.2e  |X L - I| %.2e  %.1f us" % (
        cond_shift, np.linalg.cond(A), eL, eA, eX, us))
