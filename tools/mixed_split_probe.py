import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from gogp_amd import gp as G, kernel, synth
N, D = 16384, 8
X, y = synth.make_inputs(N, D, 20251114 + 2)
base = np.log([1.0, np.sqrt(D / 6.0), 0.1])
for split in (0, 2, 1, 0, 2):
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    g.set_option("gradient_precision", 32)
    g.set_option("chain_split", split)
    g.Observe(base); g.Gradient()
    reps = 5
    t = time.perf_counter()
    for r in range(reps):
        lml = g.Observe(base + 0.01 * (r % 3)); grad = g.Gradient()
    t = (time.perf_counter() - t) / reps
    print("mixed gradient, chain_split %d: %.3f ms = %.2f evals/s lml %.12g" % (split, t * 1e3, 1 / t, lml), flush=True)
    g.close()
