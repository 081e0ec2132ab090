"""BASELINE config 4 shape on ONE GPU: Matern-5/2 (reference coefficient), N=32768, D=16."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import kernel, synth
from gogp_amd import gp as G
n, D = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 16
X, y = synth.make_inputs(n, D, 20251117)
g = G.GP(D, kernel.Scaled(kernel.Matern52), kernel.UniformNoise, X=X, Y=y)
x0 = np.log([1.0, math.sqrt(D / 6.0), 0.1])
for k in range(3):
    t = time.time(); lml = g.Observe(x0 + 0.01 * k); to = time.time() - t
    t = time.time(); gr = g.Gradient(); tg = time.time() - t
    print("N=%d D=%d matern52: lml=%.6f grad=%s observe %.1f ms gradient %.1f ms -> %.3f eval/s"
          % (n, D, lml, gr, to * 1e3, tg * 1e3, 1 / (to + tg)), flush=True)
# directional-derivative check
v = np.array([0.3, -0.5, 0.8]); v /= np.linalg.norm(v); h = 1e-4
fd = (g.Observe(x0 + h * v) - g.Observe(x0 - h * v)) / (2 * h)
g.Observe(x0); gr = g.Gradient()
print("directional derivative fd=%.6f analytic=%.6f rel=%.2e" % (fd, gr @ v, abs(fd - gr @ v) / abs(fd)))
