"""Tutorial-sized evaluations (N <= 128: option tiny, one launch for the whole factorisation) against the general sweep:
time per Observe + Gradient, per Observe, 8 candidates per launch sequence, Produce of one point; agreement of LML,
gradient, alpha, mu / sigma.  usage: python3 tools/tiny_probe.py [N,N,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gogp_amd import gp as G, kernel
Ns = [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "20,64,128".split(","))]
rng = np.random.default_rng(3)
for N in Ns:
    X = np.linspace(0, 2 * np.pi, N)[:, None]
    y = np.sin(X[:, 0]) + 0.1 * rng.normal(size=N)
    x = np.log([1.0, 0.7, 0.2])
    Z = rng.uniform(0, 6, (1, 1))
    res = {}
    for tiny in (0, 1, 0, 1):
        g = G.GP(1, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
        g.set_option("tiny", tiny)
        lml = g.Observe(x); grad = g.Gradient(); mu, sg = g.Produce(Z); alpha = g.Alpha.copy()
        reps = 200
        torch.cuda.synchronize(); t = time.perf_counter()
        for r in range(reps):
            g.Observe(x + 1e-4 * (r % 7)); g.Gradient()
        torch.cuda.synchronize(); t1 = (time.perf_counter() - t) / reps
        t = time.perf_counter()
        for r in range(reps):
            g.Observe(x + 1e-4 * (r % 7))
        torch.cuda.synchronize(); t0 = (time.perf_counter() - t) / reps
        xs = np.stack([x + 1e-3 * c for c in range(8)])
        for _ in range(3): g.observe_gradient_candidates(xs)
        torch.cuda.synchronize(); t = time.perf_counter()
        for r in range(50):
            cl, cg, cs = g.observe_gradient_candidates(xs + 1e-5 * r)
        torch.cuda.synchronize(); tc = (time.perf_counter() - t) / 50
        l0 = g.Observe(xs[0]); g0 = g.Gradient()
        cl, cg, cs = g.observe_gradient_candidates(xs)
        same = cl[0] == l0 and np.array_equal(cg[0], g0)
        res.setdefault(tiny, (lml, grad, mu, sg, alpha))
        print("N %3d tiny %d: Observe + Gradient %.1f us, Observe %.1f us, 8 candidates %.1f us (%.0f evals/s), candidates == single: %s" %
              (N, tiny, t1 * 1e6, t0 * 1e6, tc * 1e6, 8 / tc, same), flush=True)
        g.close()
    a, b = res[0], res[1]
    print("   agreement tiny vs general: lml %.2e grad %.2e mu %.2e sigma %.2e alpha %.2e" % (
        abs(a[0] - b[0]) / abs(a[0]), np.abs(a[1] - b[1]).max() / np.abs(a[1]).max(), np.abs(a[2] - b[2]).max() / np.abs(a[2]).max(),
        np.abs(a[3] - b[3]).max() / np.abs(a[3]).max(), np.abs(a[4] - b[4]).max() / np.abs(a[4]).max()))
