"""Float shards on an ill-conditioned 1-D case (kernel.Normal, ConstantNoise(0.1), n = 5651, grid 2 x 4): sigma / mu error
against the fp64 single-GPU path with the diagonal tiles in fp64 (diag_fp64 1 / 0), beside the single-GPU fp32 path with and
without the one-pass Produce for few test points."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gogp_amd import gp as G, kernel
from gogp_amd.sharded import ShardedGP
import loopback
for seed, m in ((1, 7), (2, 150)):
    rng = np.random.default_rng(seed)
    n, D = 5651, 1
    X = rng.uniform(0, 1, (n, D))
    y = np.sin(2 * np.pi * X).sum(1) + 0.1 * rng.normal(size=n); y = (y - y.mean()) / y.std()
    simil, noise = kernel.Normal, kernel.ConstantNoise(0.1)
    x = np.log([0.3 * np.exp(0.1 * rng.normal())])
    Z = rng.uniform(-0.1, 1.1, (m, D))
    ref = G.GP(D, simil, noise, X=X, Y=y)
    ref.Observe(x); mu_o, sg_o = ref.Produce(Z); ref.close()
    err = lambda mu, sg: (np.abs(mu - mu_o).max() / np.abs(mu_o).max(), np.nanmax(np.abs(sg - sg_o)) / np.nanmax(np.abs(sg_o)))
    for small in (64, 0):
        g = G.GP(D, simil, noise, X=X, Y=y, precision=32)
        g.set_option("produce_small_max", small)
        g.Observe(x); e = err(*g.Produce(Z)); g.close()
        print("M %3d single-GPU fp32, produce_small_max %2d: mu %.2e sigma %.2e" % (m, small, e[0], e[1]), flush=True)
    for d64 in (1, 0):
        def rank_fn(r, lb):
            sh = ShardedGP(D, simil, noise, X=X, Y=y, device=0, precision=32, grid=(2, 4), rank=r, world=8,
                           exchange=lb.exchange, allreduce=lb.allreduce)
            sh.set_option("diag_fp64", d64)
            sh.Observe(x); out = sh.Produce(Z); sh.close()
            return out
        outs, _ = loopback.run_ranks(8, rank_fn)
        e = err(*outs[0])
        print("M %3d float shards 2x4, diag_fp64 %d: mu %.2e sigma %.2e" % (m, d64, e[0], e[1]), flush=True)
