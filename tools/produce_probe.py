"""Wall time of gogp_produce (factor resident) against the number of independent substitution chains (option
produce_groups) and M, without profiling events on the launches.
usage: python3 tools/produce_probe.py [N] [D] ["opt=val ..."] [M,M,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
X, y = synth.make_inputs(N, D, 20251114 + 2)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
g.Observe(x); g.Gradient()
rng = np.random.default_rng(1)
opts = [o.split("=") for o in sys.argv[3].split()] if len(sys.argv) > 3 else []
for k, v in opts:
    g.set_option(k, int(v))
print("options:", opts)
for M in (int(a) for a in (sys.argv[4].split(",") if len(sys.argv) > 4 else "1,256,1024,4096".split(","))):
    Z = rng.uniform(0, 1, (M, D))
    for grp in (1, 2):
        g.set_option("produce_groups", grp)
        g.Produce(Z)
        t = time.perf_counter()
        for _ in range(5):
            g.Produce(Z)
        t = (time.perf_counter() - t) / 5
        print("M %5d groups %d: %.3f ms  (N^2 M / t = %.1f TFLOP/s)" % (M, grp, t * 1e3, float(N) * N * M / t / 1e12), flush=True)
