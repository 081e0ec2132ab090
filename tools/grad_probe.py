"""Gradient of a many-dimensional ARD kernel at a mid size: single-GPU path, sharded 1x1 path and the
CPU oracle side by side, each repeated (run-to-run determinism)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import loopback
from gogp_amd import kernel as _k, gp as G
from gogp_amd.sharded import ShardedGP
from oracle.oracle import FastOracle
D = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6904
rng = np.random.default_rng(5)
simil, noise = _k.Scaled(_k.ARD(_k.Normal, D)), _k.UniformNoise
X = rng.uniform(0, 1, (n, D))
y = np.sin(2 * np.pi * X).sum(1) / np.sqrt(D) + 0.1 * rng.normal(size=n); y = (y - y.mean()) / y.std()
x = np.log(np.array([1.1] + [2.5 + 0.03 * i for i in range(D)] + [0.2]))
o = FastOracle(D, simil, noise); o.set_data(X, y)
lo, go = o.Observe(x), o.Gradient()
print("oracle lml %.12g" % lo, flush=True)
for rep in range(3):
    g = G.GP(D, simil, noise, X=X, Y=y)
    l1, g1 = g.Observe(x), g.Gradient(); g.close()
    print("single rep %d: lml err %.2e grad err %.3e" % (rep, abs(l1 - lo) / abs(lo), np.abs(g1 - go).max() / np.abs(go).max()), flush=True)
for rep in range(3):
    def rank_fn(r, lb):
        sh = ShardedGP(D, simil, noise, X=X, Y=y, device=0, grid=(1, 1), rank=r, world=1, exchange=lb.exchange, allreduce=lb.allreduce)
        l, gr = sh.Observe(x), sh.Gradient()
        gr2 = sh.Gradient()
        sh.close(); return l, gr, gr2
    (l2, g2, g2b), = loopback.run_ranks(1, rank_fn)[0]
    print("sharded rep %d: lml err %.2e grad err %.3e, second Gradient() %.3e; worst index %d" % (
        rep, abs(l2 - lo) / abs(lo), np.abs(g2 - go).max() / np.abs(go).max(), np.abs(g2b - go).max() / np.abs(go).max(),
        int(np.argmax(np.abs(g2 - go)))), flush=True)
    if rep == 0:
        print("   relative error per parameter:", np.array2string(np.abs(g2 - go) / np.abs(go).max(), precision=1), flush=True)
