import sys
sys.path.insert(0, ".")
import numpy as np
from gogp_amd import gp as G, kernel, synth
for N in (3000, 12000):
    D = 4
    X, y = synth.make_inputs(N, D, 7)
    x = np.log([1.0, 0.7, 0.1])
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    g.set_option("kinv_fused", 0)
    g.Observe(x); g0 = g.Gradient().copy(); l0 = g.LML()
    for sp in (50, 80):
        g.set_option("kinv_split", sp)
        g.Observe(x); g1 = g.Gradient().copy()
        g.Observe(x)            # a factorisation whose first launch nobody picks up
        g.Observe(x); g2 = g.Gradient().copy()
        print(N, sp, "bit-identical:", np.array_equal(g0, g1), np.array_equal(g0, g2), np.abs(g1 - g0).max())
    g.close()
