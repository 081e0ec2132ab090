"""Quick probe for a GPU session: MFMA f64 peak, then Observe / Gradient timings."""
import math
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gogp_amd import kernel  # noqa: E402
from gogp_amd import gp as G  # noqa: E402

print("version:", G._lib.lib().gogp_version().decode(), flush=True)
for it in (2000, 20000, 100000):
    print("mfma f64 peak iters=%d: %.2f TFLOP/s, %.1f cycles/MFMA/SIMD, clock %.0f MHz"
          % ((it,) + G.mfma_f64_peak(it, details=True)), flush=True)

import os
sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096]
opts = [kv.split("=") for kv in os.environ.get("GOGP_OPTS", "").split(",") if kv]
for n in sizes:
    D = 8
    rng = np.random.default_rng(n)
    X = rng.uniform(0, 1, (n, D))
    y = np.sin(2 * np.pi * X).sum(1) / np.sqrt(D) + 0.1 * rng.normal(size=n)
    y = (y - y.mean()) / y.std()
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    x0 = np.log([1.0, math.sqrt(D / 6.0), 0.1])
    for k_, v_ in opts:
        g.set_option(k_, int(v_))
    t = time.time(); lml = g.Observe(x0); t_first = time.time() - t
    t = time.time(); gr = g.Gradient(); t_gfirst = time.time() - t
    g.profile_enable(True)
    to, tg = [], []
    for k in range(3):
        x = x0 + 0.01 * (k + 1)
        t = time.time(); lml = g.Observe(x); to.append(time.time() - t)
        t = time.time(); gr = g.Gradient(); tg.append(time.time() - t)
    ms, nl, fl, busy = g.profile_read()
    print("N=%d lml=%.6f grad=%s" % (n, lml, gr))
    print("  first: observe %.1f ms gradient %.1f ms" % (t_first * 1e3, t_gfirst * 1e3))
    print("  steady: observe %.2f ms gradient %.2f ms -> %.2f eval/s" % (
        min(to) * 1e3, min(tg) * 1e3, 1.0 / (min(to) + min(tg))))
    print("  gemm kernel: sum %.2f ms/eval, busy(union) %.2f ms/eval over %d launches/eval; "
          "N^3/busy = %.2f TFLOP/s" % (ms / 3, busy / 3, nl // 3,
                                       float(n) ** 3 / (busy / 3 * 1e-3) / 1e12 if busy > 0 else 0), flush=True)
    g.close()
