"""Does the n-th GP handle of a process run as fast as the first?  (stream -> HW queue mapping)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import kernel, synth
from gogp_amd import gp as G
n, D = 4096, 8
X, y = synth.make_inputs(n, D, 1)

def timeit(g):
    ts = []
    for k in range(5):
        t = time.time(); g.Observe(synth.log_theta_cycle(D, k)); g.Gradient(); ts.append(time.time() - t)
    return min(ts) * 1e3

mode = sys.argv[1] if len(sys.argv) > 1 else "seq"
out = []
if mode == "seq":       # create, time, destroy, repeat
    for i in range(5):
        g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
        out.append(timeit(g)); g.close()
elif mode == "live":    # all handles stay alive
    gs = [G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y) for _ in range(5)]
    out = [timeit(g) for g in gs]
    out += [timeit(g) for g in gs]
print("%s GPU_MAX_HW_QUEUES=%s: ms per eval by handle: %s" % (mode, os.environ.get("GPU_MAX_HW_QUEUES"), " ".join("%.2f" % v for v in out)), flush=True)
