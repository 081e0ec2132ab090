"""Option gradient_precision = 32 on an fp64 handle (fp64 factorisation / LML / alpha / Produce, Y = L^-T and K^-1 on the
fp32 tile kernel) against the native fp64 evaluation: time per Observe + Gradient, LML, gradient difference.
usage: python3 tools/mixed_probe.py [N] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
X, y = synth.make_inputs(N, D, 20251114 + 2)
base = np.log([1.0, np.sqrt(D / 6.0), 0.1])
res = {}
for gp_ in (64, 32, 64, 32):
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    g.set_option("gradient_precision", gp_)
    g.Observe(base); g.Gradient()
    reps = 5
    t = time.perf_counter()
    for r in range(reps):
        lml = g.Observe(base + 0.01 * (r % 3)); grad = g.Gradient()
    t = (time.perf_counter() - t) / reps
    lml = g.Observe(base); grad = g.Gradient()
    res.setdefault(gp_, (lml, grad))
    print("gradient_precision %d: %.3f ms per Observe + Gradient = %.2f evals/s; lml %.9f grad %s" % (gp_, t * 1e3, 1 / t, lml, grad), flush=True)
    g.close()
l64, g64 = res[64]; l32, g32 = res[32]
print("LML identical: %s; gradient max |diff| / max |g| = %.3e" % (l64 == l32, np.abs(g32 - g64).max() / np.abs(g64).max()))
