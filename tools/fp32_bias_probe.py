"""Where the fp32 path's error on the ill-conditioned golden case comes from: is the float factor's error random (rounding
to nearest in the fp32 MFMA accumulation) or biased?  Signed mean against RMS of the element errors of L (fp32 handle
against fp64 handle), of K^-1's trace, and the same with the fp64 factor + float inverse (gradient_precision = 32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gogp_amd import gp as G, kernel
d = np.load(os.path.join(ROOT, "tests", "golden", "fp32_illcond_matern32.npz"))
X, y, x = d["X"], d["y"], d["x"]
simil, noise = kernel.Scaled(kernel.Matern32), kernel.ScaledNoise(0.01)  # tests/cases.py: "matern32"
g64 = G.GP(2, simil, noise, X=X, Y=y)
g64.Observe(x)
grad64 = g64.Gradient()
L64 = g64.L.copy()
n = len(y)
Kinv = np.linalg.inv(L64 @ L64.T)
g32 = G.GP(2, simil, noise, X=X, Y=y, precision=32)
if len(sys.argv) > 1:
    g32.set_option("diag_fp64", int(sys.argv[1]))
g32.Observe(x)
grad32 = g32.Gradient()
L32 = g32.L.copy()
il = np.tril_indices(n)
e = (L32 - L64)[il]
ref = np.abs(L64[il]).mean()
print("L: mean signed error %.3e, rms error %.3e (of mean |L| = %.3e); diagonal: mean signed %.3e rms %.3e" %
      (e.mean() / ref, np.sqrt((e ** 2).mean()) / ref, ref, (np.diag(L32) - np.diag(L64)).mean() / np.diag(L64).mean(),
       np.sqrt(((np.diag(L32) - np.diag(L64)) ** 2).mean()) / np.diag(L64).mean()))
for b0 in range(0, n, 256):
    sl = slice(b0, min(n, b0 + 256))
    dd = (np.diag(L32) - np.diag(L64))[sl] / np.diag(L64)[sl]
    print("  diagonal block %d: mean signed rel. error of L_ii %.3e, rms %.3e" % (b0 // 256, dd.mean(), np.sqrt((dd ** 2).mean())))
Y32 = np.linalg.inv(L32.astype(np.float64))   # the exact inverse of the float factor
print("tr(K^-1): exact %.9f; from the float factor inverted exactly %.9f (err %.3e)" %
      (np.trace(Kinv), (Y32 ** 2).sum(), (Y32 ** 2).sum() - np.trace(Kinv)))
scale = np.abs(grad64).max()
print("gradient fp32 - fp64, of the largest component:", (grad32 - grad64) / scale)
g64.set_option("gradient_precision", 32)
g64.Observe(x)
print("gradient (fp64 factor, float inverse) - fp64:", (g64.Gradient() - grad64) / scale)
