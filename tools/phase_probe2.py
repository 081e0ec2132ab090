import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G, kernel, synth
N, D = 16384, 8
X, y = synth.make_inputs(N, D, 20251114 + 2)
base = np.log([1.0, np.sqrt(D / 6.0), 0.1])
for grp in sys.argv[1:]:
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    for o in grp.split():
        g.set_option(o.split("=")[0], int(o.split("=")[1]))
    g.Observe(base); g.Gradient()
    tO = tG = 0.0
    reps = 5
    for r in range(reps):
        t0 = time.perf_counter(); lml = g.Observe(base + 0.01 * (r % 3)); t1 = time.perf_counter(); grad = g.Gradient(); t2 = time.perf_counter()
        tO += t1 - t0; tG += t2 - t1
    print("%-60s Observe %.2f ms + Gradient %.2f ms = %.2f ms (%.2f evals/s)" % (grp, tO / reps * 1e3, tG / reps * 1e3, (tO + tG) / reps * 1e3, reps / (tO + tG)), flush=True)
    g.close()
