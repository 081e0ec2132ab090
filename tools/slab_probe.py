"""Option chain_slabs (panel128.hip: 64-row slabs per workgroup of the one-launch chain step; the result does not depend on
it): Observe only (eager = 0) and Observe + Gradient with chain_split = 2 at a large N, alternating.
usage: python3 tools/slab_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = 8
X, y = synth.make_inputs(N, D, 20251114 + 2)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
g.set_option("chain_split", 2)
for slabs in (1, 2, 4, 1, 2, 4):
    g.set_option("chain_slabs", slabs)
    out = []
    for eager in (0, 1):
        g.set_option("eager", eager)
        lml = g.Observe(x)
        if eager: g.Gradient()
        reps = 6
        torch.cuda.synchronize()
        t = time.perf_counter()
        for r in range(reps):
            g.Observe(x + 1e-3 * r)
            if eager: g.Gradient()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / reps * 1e3)
    print("N %d chain_slabs %d: Observe only %.3f ms, Observe + Gradient %.3f ms (lml %.12g)" % (N, slabs, out[0], out[1], lml), flush=True)
g.close()
