"""A/B of the chain server (resident diagonal-block workgroup) against one launch per block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import kernel, synth
from gogp_amd import gp as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = 8
X, y = synth.make_inputs(n, D, 20251116)
for mode in (0, 1, 0, 1):
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    g.set_option("chain_server", mode)
    ts = []
    for k in range(6):
        t = time.time()
        lml = g.Observe(synth.log_theta_cycle(D, k))
        gr = g.Gradient()
        ts.append(time.time() - t)
    print("chain_server=%d n=%d: lml=%.9f grad=%s  best %.2f ms  median %.2f ms" % (
        mode, n, lml, np.array2string(gr, precision=6), min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3), flush=True)
    g.close()
