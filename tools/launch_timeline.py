"""Throughput timeline of ONE Observe+Gradient from the library's own per-launch HIP events
(gogp_profile_read_launches): flops of every tile-kernel launch spread uniformly over its interval,
binned per millisecond; per kernel class (mode, K) the summed flops, summed duration and the rate while
it was the only class... usage: python3 tools/launch_timeline.py [config] [NAME=VALUE options...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import configs, gp as G

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = configs.workload(cfg, None, None)
X, y = wl.inputs()
g = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y)
for ov in sys.argv[2:]:
    g.set_option(ov.split("=")[0], int(ov.split("=")[1]))
for k in range(2):
    g.Observe(wl.log_theta(k)); g.Gradient()
g.profile_enable(True)
t = time.perf_counter()
g.Observe(wl.log_theta(5)); g.Gradient()
wall = (time.perf_counter() - t) * 1e3
t0, t1, fl, tag = g.profile_read_launches()
end = t1.max()
print("wall %.2f ms, last launch ends at %.2f ms, %d launches, %.3f TFLOP launched (N^3 = %.3f)" % (
    wall, end, len(t0), fl.sum() / 1e12, float(wl.N) ** 3 / 1e12))
nb = int(np.ceil(end))
bins = np.zeros(nb)
for a, b, f in zip(t0, t1, fl):
    d = max(b - a, 1e-6)
    i0, i1 = int(a), min(int(b), nb - 1)
    for i in range(i0, i1 + 1):
        ov = min(b, i + 1) - max(a, i)
        if ov > 0:
            bins[i] += f * ov / d
print("TFLOP/s per ms bin:")
for i in range(0, nb, 10):
    print("  %3d ms: %s" % (i, " ".join("%5.1f" % (v / 1e9) for v in bins[i:i + 10])))
cls = {}
for a, b, f, tg in zip(t0, t1, fl, tag):
    mode, K, tiles = tg // 100000000, (tg // 100000) % 1000 * 16, tg % 100000
    size = "small" if tiles < 384 else ("mid" if tiles < 3072 else "big")
    key = (int(mode), int(K), size)
    c = cls.setdefault(key, [0, 0.0, 0.0])
    c[0] += 1; c[1] += f; c[2] += b - a
print("class (mode, K, size): launches, GFLOP, summed duration ms, GFLOP / summed ms")
for k, c in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print("  %-22s %4d  %9.1f  %8.2f  %6.1f" % (k, c[0], c[1] / 1e9, c[2], c[1] / 1e9 / max(c[2], 1e-9)))
# idle time of the two bulk streams: the Cholesky trailing updates (mode 1, K >= 512, not small) run in order on
# the main stream, the inverse's big updates (mode 0, K >= 512, not small) on s2 -- the gap between one such
# launch's end and the next one's start is time that stream waited for the chain
for name, sel in (("Cholesky bulk (LOWER)", lambda m, K, t: m == 1 and K >= 512 and t >= 384),
                  ("inverse bulk (RECT big/mid)", lambda m, K, t: m == 0 and K >= 512 and t >= 384)):
    iv = sorted((a, b) for a, b, tg in zip(t0, t1, tag)
                if sel(tg // 100000000, (tg // 100000) % 1000 * 16, tg % 100000))
    gaps = [(iv[i + 1][0] - iv[i][1], iv[i][1]) for i in range(len(iv) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print("%s: %d launches, busy %.2f ms, idle between them %.2f ms in %d gaps; largest: %s" % (
        name, len(iv), sum(b - a for a, b in iv), sum(g for g, _ in pos), len(pos),
        ", ".join("%.2f ms at t=%.1f" % g for g in sorted(pos, reverse=True)[:6])))
g.close()
