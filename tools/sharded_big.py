"""One sharded evaluation at a BASELINE size on a Pr x Pc grid of rank THREADS sharing one GPU (callback
transport through queues): correctness at realistic block counts, against the single-GPU path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import loopback
from gogp_amd import configs
from gogp_amd import gp as G
from gogp_amd.sharded import ShardedGP
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
grid = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (2, 4)
nobs = int(sys.argv[3]) if len(sys.argv) > 3 else None
prec = int(sys.argv[4]) if len(sys.argv) > 4 else 64  # 32: float tiles on the shards (reference below stays fp64)
wl = configs.workload(cfg, nobs)
X, y = wl.inputs()
x = wl.log_theta(0)
Z = wl.test_points(64)
ref = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0)
lml_o, grad_o = ref.Observe(x), ref.Gradient()
mu_o, sg_o = ref.Produce(Z)
alpha_o = ref.Alpha
ref.close()
if prec == 32:  # what float matrices cost on ONE GPU at this size, for comparison
    r32 = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0, precision=32)
    l32, g32 = r32.Observe(x), r32.Gradient()
    m32, s32 = r32.Produce(Z)
    a32 = r32.Alpha
    r32.close()
    print("single GPU fp32 vs fp64: lml %.1e grad %.1e mu %.1e sigma %.1e alpha %.1e" % (
        abs(l32 - lml_o) / abs(lml_o), np.abs(g32 - grad_o).max() / max(1.0, np.abs(grad_o).max()),
        np.abs(m32 - mu_o).max() / np.abs(mu_o).max(), np.abs(s32 - sg_o).max() / np.abs(sg_o).max(),
        np.abs(a32 - alpha_o).max() / np.abs(alpha_o).max()), flush=True)
world = grid[0] * grid[1]

def rank_fn(r, lb):
    sh = ShardedGP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0, precision=prec, grid=grid, rank=r, world=world,
                   exchange=lb.exchange, allreduce=lb.allreduce)
    t0 = time.time()
    lml = sh.Observe(x)
    grad = sh.Gradient()
    dt = time.time() - t0
    mu, sg = sh.Produce(Z)
    out = (lml, grad, mu, sg, sh.Alpha, sh.local_bytes(), dt)
    sh.close()
    return out

outs, lb = loopback.run_ranks(world, rank_fn)
for r, (lml, grad, mu, sg, al, nb, dt) in enumerate(outs):
    e = (abs(lml - lml_o) / abs(lml_o), np.abs(grad - grad_o).max() / max(1.0, np.abs(grad_o).max()),
         np.abs(mu - mu_o).max() / np.abs(mu_o).max(), np.abs(sg - sg_o).max() / np.abs(sg_o).max(),
         np.abs(al - alpha_o).max() / np.abs(alpha_o).max())
    tol = (1e-10, 1e-8, 1e-8, 1e-7, 1e-8) if prec == 64 else (5e-6, 1e-3, 1e-2, 5e-2, 1e-3)  # fp32: gross-error bounds
    assert all(v < t for v, t in zip(e, tol)), (r, e)
    if r == 0:
        print("config %d N=%d grid %dx%d, tiles fp%d: rel. errors vs single GPU (fp64): lml %.1e grad %.1e mu %.1e sigma %.1e alpha %.1e; "
              "%.2f GB per rank; %.2f GB exchanged per rank; %.1f s (host-synchronous rehearsal transport)" % (
                  cfg, wl.N, grid[0], grid[1], prec, *e, nb / 1e9, sum(lb.sent_bytes) / world / 1e9, dt), flush=True)
print("ok")
