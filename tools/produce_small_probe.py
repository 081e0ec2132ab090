"""Produce for few test points: the one-pass substitution kernel (trsm_small.hip, option produce_small_max) against the
tile-kernel route, wall time per call (factor resident, GPU otherwise idle) and agreement.
usage: python3 tools/produce_small_probe.py [N] [D] [M,M,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Ms = [int(a) for a in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,8,16,17,32,33,64".split(","))]
X, y = synth.make_inputs(N, D, 20251114 + 2)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
g.Observe(x); g.Gradient()
rng = np.random.default_rng(1)
tri_bytes = 8.0 * N * N / 2
for M in Ms:
    Z = rng.uniform(0, 1, (M, D))
    res = {}
    for name, mx in (("one-pass", 64), ("tile-kernel", 0)):
        g.set_option("produce_small_max", mx)
        res[name] = g.Produce(Z)
        reps = 10
        t = time.perf_counter()
        for _ in range(reps):
            g.Produce(Z)
        t = (time.perf_counter() - t) / reps
        print("N %d M %3d %-11s %.3f ms per call  (8 N^2 / 2 bytes / t = %.2f TB/s)" % (N, M, name, t * 1e3, tri_bytes / t / 1e12),
              flush=True)
    dmu = np.abs(res["one-pass"][0] - res["tile-kernel"][0]).max() / max(1e-300, np.abs(res["tile-kernel"][0]).max())
    dsg = np.abs(res["one-pass"][1] - res["tile-kernel"][1]).max() / max(1e-300, np.abs(res["tile-kernel"][1]).max())
    print("      agreement: mu %.2e sigma %.2e" % (dmu, dsg), flush=True)
# the harness's order: Observe (gradient preparation running), then ONE point
g.set_option("produce_small_max", 64)
Z = rng.uniform(0, 1, (1, D))
for _ in range(2):
    t = time.perf_counter()
    g.Observe(x)
    t1 = time.perf_counter()
    g.Produce(Z)
    t2 = time.perf_counter()
    print("Observe %.2f ms, then Produce(1 point) behind it %.2f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3), flush=True)
