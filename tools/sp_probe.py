"""Observe only (eager = 0) at a large N against the super-panel options (superpanel, superpanel_head, head_remaining) and
the chain form: which blocking the factorisation alone wants (the defaults were tuned with the inverse beside it).
usage: python3 tools/sp_probe.py [N]"""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = 8
X, y = synth.make_inputs(N, D, 20251114 + 2)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
g.set_option("eager", 0)
combos = [(2, 3, 16), (2, 0, 16), (3, 0, 16), (4, 0, 16), (2, 4, 16), (2, 4, 32), (2, 6, 32), (3, 4, 24), (4, 6, 32), (2, 3, 32), (1, 3, 16), (2, 2, 0)]
for split in (2, 0):
    g.set_option("chain_split", split)
    for sp, head, rem in combos:
        g.set_option("superpanel", sp); g.set_option("superpanel_head", head); g.set_option("head_remaining", rem)
        lml = g.Observe(x)
        reps = 6
        torch.cuda.synchronize()
        t = time.perf_counter()
        for r in range(reps):
            g.Observe(x + 1e-3 * r)
        torch.cuda.synchronize()
        print("N %d chain_split %d superpanel %d head %d remaining %2d: Observe only %.3f ms (lml %.12g)" %
              (N, split, sp, head, rem, (time.perf_counter() - t) / reps * 1e3, lml), flush=True)
g.close()
