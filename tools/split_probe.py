"""Option chain_split (the diagonal block in two 128-halves, products on the tile kernel) against the 256-block kernel:
one Observe + Gradient at a time, Observe only (eager = 0), 8 candidates per launch sequence.
usage: python3 tools/split_probe.py [N,N,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gogp_amd import gp as G, kernel, synth
Ns = [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1024,2048,4096,8192,16384".split(","))]
for N in Ns:
    D = 4 if N <= 4096 else 8
    X, y = synth.make_inputs(N, D, 20251114 + 1)
    x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    res = {}
    for split in (0, 1, 2, 0, 1, 2):
        g.set_option("chain_split", split)
        g.set_option("eager", 1)
        lml = g.Observe(x); g.Gradient()
        reps = 20 if N <= 4096 else 6
        torch.cuda.synchronize()
        t = time.perf_counter()
        for r in range(reps):
            g.Observe(x + 1e-3 * r); g.Gradient()
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t) / reps
        g.set_option("eager", 0)
        g.Observe(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for r in range(reps):
            g.Observe(x + 1e-3 * r)
        torch.cuda.synchronize()
        t0 = (time.perf_counter() - t) / reps
        g.set_option("eager", 1)
        tc = float("nan")
        if N <= 8192:
            xs = np.stack([x + 1e-3 * c for c in range(8)])
            g.observe_gradient_candidates(xs); g.observe_gradient_candidates(xs)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for r in range(max(3, reps // 4)):
                g.observe_gradient_candidates(xs + 1e-4 * r)
            torch.cuda.synchronize()
            tc = (time.perf_counter() - t) / max(3, reps // 4)
        print("N %5d chain_split %d: Observe+Gradient %.3f ms, Observe only %.3f ms, 8 candidates %.3f ms  (lml %.12g)" %
              (N, split, t1 * 1e3, t0 * 1e3, tc * 1e3, lml), flush=True)
    g.close()
