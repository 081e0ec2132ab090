"""Throughput of k concurrent Observe+Gradient evaluations on one GPU (gogp_observe_gradient_batch)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from gogp_amd import configs
from gogp_amd import gp as G
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nobs = int(sys.argv[2]) if len(sys.argv) > 2 else None
wl = configs.workload(cfg, nobs)
X, y = wl.inputs()
for k in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,2,3,4,6,8".split(","))]:
    gps = [G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0) for _ in range(k)]
    xs = np.array([wl.log_theta(i) for i in range(k)])
    G.observe_gradient_batch(gps, xs)
    reps = 20
    t0 = time.perf_counter()
    for r in range(reps):
        xs = np.array([wl.log_theta(r * k + i) for i in range(k)])
        lmls, grads = G.observe_gradient_batch(gps, xs)
    dt = time.perf_counter() - t0
    print("N=%d k=%d: %.1f evals/s (%.2f ms per batch)" % (wl.N, k, reps * k / dt, dt / reps * 1e3), flush=True)
    for g in gps:
        g.close()
