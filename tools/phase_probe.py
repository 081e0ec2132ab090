"""Wall time of Observe and of Gradient at config 3, with the triangular inverse fused into the factorisation
(eager = 1, default) and deferred to Gradient (eager = 0).   usage: python3 tools/phase_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gogp_amd import configs, gp as G
wl = configs.workload(3, None, None)
X, y = wl.inputs()
for eager in (1, 0):
    g = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y)
    g.set_option("eager", eager)
    for k in range(2):
        g.Observe(wl.log_theta(k)); g.Gradient()
    to, tg = [], []
    for k in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.Observe(wl.log_theta(k)); 
        if eager == 0: torch.cuda.synchronize()
        t1 = time.perf_counter(); g.Gradient(); torch.cuda.synchronize(); t2 = time.perf_counter()
        to.append((t1 - t0) * 1e3); tg.append((t2 - t1) * 1e3)
    print("eager=%d  Observe %.2f ms  Gradient %.2f ms  total %.2f" % (eager, np.median(to), np.median(tg), np.median(to) + np.median(tg)))
    g.close()
