"""BASELINE config 5 shape on ONE GPU in fp64: ARD-RBF, N=65536, D=32 (P = 34)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import kernel, synth
from gogp_amd import gp as G
n, D = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 32
X, y = synth.make_inputs(n, D, 20251118)
g = G.GP(D, kernel.Scaled(kernel.ARD(kernel.Normal, D)), kernel.UniformNoise, X=X, Y=y)
ls = math.sqrt(D / 6.0) * (1 + np.arange(D) / (2.0 * D))
x0 = np.log(np.concatenate([[1.0], ls, [0.1]]))
for k in range(2):
    t = time.time(); lml = g.Observe(x0 + 0.01 * k); to = time.time() - t
    t = time.time(); gr = g.Gradient(); tg = time.time() - t
    print("N=%d D=%d ARD: lml=%.6f |grad|inf=%.4g observe %.1f ms gradient %.1f ms -> %.4f eval/s (N^3/t = %.1f TFLOP/s)"
          % (n, D, lml, np.abs(gr).max(), to * 1e3, tg * 1e3, 1 / (to + tg), float(n) ** 3 / (to + tg) / 1e12), flush=True)
v = np.random.default_rng(0).normal(size=x0.size); v /= np.linalg.norm(v); h = 1e-4
fd = (g.Observe(x0 + h * v) - g.Observe(x0 - h * v)) / (2 * h)
g.Observe(x0); gr = g.Gradient()
print("directional derivative fd=%.6f analytic=%.6f rel=%.2e" % (fd, gr @ v, abs(fd - gr @ v) / abs(fd)))
