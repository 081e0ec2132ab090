"""fp32 path (option "precision" = 32) against the fp64 path: errors and time per evaluation."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from gogp_amd import configs
from gogp_amd import gp as G
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4096, 16384]
for n in sizes:
    wl = configs.workload(cfg, n)
    X, y = wl.inputs()
    Z = wl.test_points(256)
    res = {}
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for prec in (64, 32):
        g = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0, precision=prec)
        if prec == 32:
            g.set_option("refine_steps", steps)
        lml = g.Observe(wl.log_theta(0)); grad = g.Gradient()
        t0 = time.perf_counter()
        for k in range(3):
            lml = g.Observe(wl.log_theta(k + 1)); grad = g.Gradient()
        dt = (time.perf_counter() - t0) / 3
        mu, sg = g.Produce(Z)
        res[prec] = (lml, grad, mu, sg, g.Alpha, dt)
        g.close()
    a, b = res[64], res[32]
    print("config %d N=%d refine=%d: fp64 %.2f ms  fp32 %.2f ms (x%.2f)" % (cfg, n, steps, a[5] * 1e3, b[5] * 1e3, a[5] / b[5]))
    print("   lml %.9f vs %.9f  rel %.2e" % (a[0], b[0], abs(a[0] - b[0]) / abs(a[0])))
    print("   grad max rel (to max |g|) %.2e   per-component rel max %.2e" % (
        np.abs(a[1] - b[1]).max() / np.abs(a[1]).max(), (np.abs(a[1] - b[1]) / np.maximum(np.abs(a[1]), 1e-300)).max()))
    print("   mu rel %.2e  sigma rel %.2e  alpha rel %.2e" % (
        np.abs(a[2] - b[2]).max() / np.abs(a[2]).max(), np.abs(a[3] - b[3]).max() / np.abs(a[3]).max(),
        np.abs(a[4] - b[4]).max() / np.abs(a[4]).max()), flush=True)

# where it breaks: cond(K) ~ N c / sigma^2 against 1 / eps_f32 = 1.7e7
if len(sys.argv) > 4:
    n = 4096
    wl = configs.workload(cfg, n)
    X, y = wl.inputs()
    for sigma in (0.1, 0.03, 0.01, 0.003, 0.001):
        th = wl.theta.copy()
        th[-1] = sigma
        x = np.log(th)
        g64 = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0)
        l64, gr64 = g64.Observe(x), g64.Gradient()
        for steps in (1, 3):
            g32 = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0, precision=32)
            g32.set_option("refine_steps", steps)
            try:
                l32, gr32 = g32.Observe(x), g32.Gradient()
                print("sigma=%g N c/sigma^2=%.1e refine=%d: lml rel %.2e grad rel %.2e alpha rel %.2e" % (
                    sigma, n * th[0] / sigma ** 2, steps, abs(l32 - l64) / abs(l64),
                    np.abs(gr32 - gr64).max() / np.abs(gr64).max(),
                    np.abs(g32.Alpha - g64.Alpha).max() / np.abs(g64.Alpha).max()), flush=True)
            except Exception as e:
                print("sigma=%g refine=%d: %r" % (sigma, steps, e), flush=True)
            g32.close()
        g64.close()
