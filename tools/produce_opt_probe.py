"""gogp_produce at M = 1024 against the blocking options of its substitution (produce_panels, produce_groups,
produce_small_below), factor resident.  usage: python3 tools/produce_opt_probe.py [N] [M]"""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
D = 8
X, y = synth.make_inputs(N, D, 20251114 + 2)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
g.Observe(x); g.Gradient()
Z = np.random.default_rng(1).uniform(0, 1, (M, D))
for rep in range(2):
    for panels, groups, small in itertools.product((2, 3, 4, 6), (2, 4), (384, 1024)):
        g.set_option("produce_panels", panels); g.set_option("produce_groups", groups); g.set_option("produce_small_below", small)
        g.Produce(Z); g.Produce(Z)
        t = time.perf_counter()
        for _ in range(5):
            g.Produce(Z)
        t = (time.perf_counter() - t) / 5
        print("N %d M %d panels %d groups %d small_below %4d: %.3f ms = %.3f of the roof" % (N, M, panels, groups, small, t * 1e3, float(N) * N * M / t / 78.6e12), flush=True)
g.close()
