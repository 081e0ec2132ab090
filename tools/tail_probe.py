"""Option chain_tail (chain_split = -1 above N = 8192 beside the inverse: the one-launch chain step for the super-panels
with at most that many rows left): Observe + Gradient against the threshold, alternating.
usage: python3 tools/tail_probe.py [N] [tail,tail,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
tails = [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,2048,4096,6144,8192,0,2048,4096,6144,8192".split(","))]
D = 8 if N <= 16384 else 16
X, y = synth.make_inputs(N, D, 20251114 + 2)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
for tail in tails:
    g.set_option("chain_tail", tail)
    lml = g.Observe(x); g.Gradient()
    reps = 8 if N <= 16384 else 3
    torch.cuda.synchronize()
    t = time.perf_counter()
    for r in range(reps):
        g.Observe(x + 1e-3 * r); g.Gradient()
    torch.cuda.synchronize()
    print("N %d chain_tail %5d: %.3f ms per Observe + Gradient (lml %.12g)" % (N, tail, (time.perf_counter() - t) / reps * 1e3, lml), flush=True)
g.close()
