"""Throughput of k candidates per launch sequence (gogp_observe_gradient_candidates) on one GPU."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from gogp_amd import configs
from gogp_amd import gp as G
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nobs = int(sys.argv[2]) if len(sys.argv) > 2 else None
wl = configs.workload(cfg, nobs)
X, y = wl.inputs()
g = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0)
import os
if os.environ.get("CAND_SUPERPANEL"):
    g.set_option("superpanel", int(os.environ["CAND_SUPERPANEL"]))
    print("superpanel", os.environ["CAND_SUPERPANEL"], flush=True)
if os.environ.get("CAND_GRAPH"):
    g.set_option("graph", 1)
    print("hipGraph replay on", flush=True)
x0 = wl.log_theta(0)
t0 = time.perf_counter(); reps = 10
g.Observe(x0); g.Gradient()
t0 = time.perf_counter()
for r in range(reps):
    g.Observe(wl.log_theta(r)); g.Gradient()
dt1 = (time.perf_counter() - t0) / reps
print("N=%d single Observe+Gradient: %.2f ms (%.1f evals/s)" % (wl.N, dt1 * 1e3, 1 / dt1), flush=True)
for k in [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,2,4,8,16".split(","))]:
    xs = np.array([wl.log_theta(i) for i in range(k)])
    lmls, grads, st = g.observe_gradient_candidates(xs)
    ref = [(g.Observe(x), g.Gradient()) for x in xs[:2]]
    ok = all(lmls[i] == ref[i][0] and np.array_equal(grads[i], ref[i][1]) for i in range(min(k, 2)))
    reps = max(3, 24 // k)
    t0 = time.perf_counter()
    for r in range(reps):
        xs = np.array([wl.log_theta(r * k + i) for i in range(k)])
        g.observe_gradient_candidates(xs)
    dt = (time.perf_counter() - t0) / reps
    print("N=%d k=%d: %.1f evals/s (%.2f ms per batch, %.2f ms per candidate); bit-equal to single calls: %s" % (
        wl.N, k, k / dt, dt * 1e3, dt / k * 1e3, ok), flush=True)
g.close()
