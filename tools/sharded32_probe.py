"""Sharded fp32 evaluation on rank threads: errors against the fp64 oracle per grid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import loopback
from gogp_amd import configs, gp as G
from gogp_amd.sharded import ShardedGP
from oracle.oracle import FastOracle
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
wl = configs.workload(cfg, n)
X, y = wl.inputs(); Z = wl.test_points(32); x = wl.log_theta(0)
o = FastOracle(wl.D, wl.simil, wl.noise); o.set_data(X, y)
lml_o, grad_o = o.Observe(x), o.Gradient(); mu_o, sg_o = o.Produce(Z)
for prec in (64, 32):
    for grid in [(1, 1), (1, 2), (2, 2), (2, 4)]:
        world = grid[0] * grid[1]
        def rank_fn(r, lb):
            sh = ShardedGP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0, precision=prec, grid=grid, rank=r, world=world,
                           exchange=lb.exchange, allreduce=lb.allreduce)
            lml, grad = sh.Observe(x), sh.Gradient()
            mu, sg = sh.Produce(Z)
            a = sh.Alpha
            sh.close()
            return lml, grad, mu, sg, a
        outs, _ = loopback.run_ranks(world, rank_fn)
        lml, grad, mu, sg, a = outs[0]
        print("prec %d grid %dx%d: lml %.2e grad %.2e (worst idx %d) alpha %.2e mu %.2e sigma %.2e" % (
            prec, grid[0], grid[1], abs(lml - lml_o) / abs(lml_o), np.abs(grad - grad_o).max() / np.abs(grad_o).max(),
            int(np.argmax(np.abs(grad - grad_o))), np.abs(a - o.Alpha).max() / np.abs(o.Alpha).max(),
            np.abs(mu - mu_o).max() / np.abs(mu_o).max(), np.abs(sg - sg_o).max() / np.abs(sg_o).max()), flush=True)
