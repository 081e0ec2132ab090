"""One Observe + Gradient at a time at a chain-bound size against the super-panel width and the K^-1 fusion options.
usage: python3 tools/sp_small_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = 4
X, y = synth.make_inputs(N, D, 20251114 + 1)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
for rep in range(2):
    for opts in ({}, {"superpanel": 1}, {"superpanel": 3}, {"superpanel": 4}, {"kinv_fused": 0}, {"chain_prio": 0}, {"chain_prio": 2},
                 {"lookahead": 0}):
        for k, v in {"superpanel": 2, "kinv_fused": -1, "chain_prio": -1, "lookahead": 1}.items():
            g.set_option(k, v)
        for k, v in opts.items():
            g.set_option(k, v)
        lml = g.Observe(x); g.Gradient()
        reps = 20
        torch.cuda.synchronize()
        t = time.perf_counter()
        for r in range(reps):
            g.Observe(x + 1e-3 * r); g.Gradient()
        torch.cuda.synchronize()
        print("N %d %-22s: %.3f ms per Observe + Gradient" % (N, opts, (time.perf_counter() - t) / reps * 1e3), flush=True)
g.close()
