"""Aggregate throughput of k independent evaluations running concurrently on ONE GPU
(k handles, k host threads) against one at a time."""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import kernel, synth
from gogp_amd import gp as G
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D, steps = 8, 6
X, y = synth.make_inputs(n, D, 20251116)
def worker(g, rank, out):
    for k in range(steps):
        lml = g.Observe(synth.log_theta_cycle(D, k, rank)); g.Gradient()
    out.append(lml)
for nh in (1, 2, 3):
    gs = [G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y) for _ in range(nh)]
    for g in gs:  # warm-up (allocation, first-call effects)
        g.Observe(synth.log_theta_cycle(D, 0)); g.Gradient()
    out = []
    ths = [threading.Thread(target=worker, args=(g, r, out)) for r, g in enumerate(gs)]
    t = time.time(); [th.start() for th in ths]; [th.join() for th in ths]; dt = time.time() - t
    print("N=%d, %d concurrent handle(s): %.2f evals/s aggregate (%.1f ms per evaluation per handle)" % (
        n, nh, nh * steps / dt, dt / steps * 1e3), flush=True)
    for g in gs: g.close()
