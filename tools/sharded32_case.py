"""A test-matrix kernel family on the sharded fp32 path vs the single-GPU fp32 path vs the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import loopback
from cases import CASES
from gogp_amd import gp as G
from gogp_amd.sharded import ShardedGP
from oracle.oracle import FastOracle
want = sys.argv[1] if len(sys.argv) > 1 else "matern32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3030
name, D, simil, noise, ts, tn = [c for c in CASES if c[0] == want][0]
rng = np.random.default_rng(3)
X = rng.uniform(0, 1, (n, D))
y = np.sin(2 * np.pi * X).sum(1) / np.sqrt(D) + 0.1 * rng.normal(size=n); y = (y - y.mean()) / y.std()
x = np.log(np.array(list(ts) + list(tn)))
o = FastOracle(D, simil, noise); o.set_data(X, y)
lml_o, grad_o = o.Observe(x), o.Gradient()
def err(lml, grad):
    return abs(lml - lml_o) / abs(lml_o), np.abs(grad - grad_o).max() / max(1.0, np.abs(grad_o).max())
g = G.GP(D, simil, noise, X=X, Y=y, precision=32)
print("single GPU fp32: lml %.2e grad %.2e" % err(g.Observe(x), g.Gradient()), flush=True)
g.close()
for grid in [(1, 1), (1, 2), (1, 3), (2, 2), (2, 4)]:
    world = grid[0] * grid[1]
    def rank_fn(r, lb):
        sh = ShardedGP(D, simil, noise, X=X, Y=y, device=0, precision=32, grid=grid, rank=r, world=world,
                       exchange=lb.exchange, allreduce=lb.allreduce)
        out = (sh.Observe(x), sh.Gradient())
        out2 = (sh.Observe(x), sh.Gradient())
        sh.close()
        return out, out2
    outs, _ = loopback.run_ranks(world, rank_fn)
    (a, b) = outs[0]
    print("sharded fp32 %dx%d: lml %.2e grad %.2e | second evaluation: lml %.2e grad %.2e" % (grid + err(*a) + err(*b)), flush=True)
