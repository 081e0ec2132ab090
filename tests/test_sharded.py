"""One evaluation sharded over several ranks (gogp_amd.sharded.ShardedGP).

The communication layer is torch.distributed; here the ranks share ONE GPU and talk
over gloo (RCCL refuses duplicate devices), which exercises the whole ownership /
pack / broadcast / unpack / filtered-update / all-reduce logic.  Results must agree
with the single-GPU path to rounding."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from gogp_amd import kernel, synth
from gogp_amd import gp as G
from gogp_amd.sharded import ShardedGP
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ok = True
for (n, D, simil) in [(700, 3, kernel.Scaled(kernel.Normal)), (1500, 2, kernel.Scaled(kernel.Matern52)),
                      (2300, 4, kernel.Scaled(kernel.ARD(kernel.Normal, 4))),
                      (5000, 2, kernel.Scaled(kernel.Normal))]:
    X, y = synth.make_inputs(n, D, 1234 + n)
    nth = simil.NTheta() + 1
    x = np.log(np.linspace(0.6, 1.2, nth))
    x[-1] = np.log(0.2)
    ref = G.GP(D, simil, kernel.UniformNoise, X=X, Y=y, device=0)
    lml_ref = ref.Observe(x)
    grad_ref = ref.Gradient()
    sh = ShardedGP(D, simil, kernel.UniformNoise, X=X, Y=y, device=0)
    for rep in range(2):
        lml = sh.Observe(x)
        grad = sh.Gradient()
        assert abs(lml - lml_ref) <= 1e-10 * abs(lml_ref), (rank, n, lml, lml_ref)
        assert np.abs(grad - grad_ref).max() <= 1e-8 * max(1.0, np.abs(grad_ref).max()), (rank, n, grad, grad_ref)
    np.testing.assert_allclose(sh.Alpha, ref.Alpha, rtol=1e-8, atol=1e-10)
    Z = synth.make_test_points(9, D, 5)
    mu, sg = sh.Produce(Z)
    mu_r, sg_r = ref.Produce(Z)
    np.testing.assert_allclose(mu, mu_r, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(sg, sg_r, rtol=1e-8, atol=1e-10)
    # Absorb path (no gradient) is sharded too
    sh.ThetaSimil, sh.ThetaNoise = list(np.exp(x[:-1])), [float(np.exp(x[-1]))]
    sh.Absorb(X, y)
    assert abs(sh.LML() - lml_ref) <= 1e-10 * abs(lml_ref)
# not positive definite: every rank must see the error
Xd = np.array([[0.0], [0.0], [1.0]]); yd = np.array([1.0, 1.0, 0.0])
bad = ShardedGP(1, kernel.Normal, kernel.ConstantNoise(0.0), ThetaSimil=[1.0], device=0)
try:
    bad.Absorb(Xd, yd)
    raise SystemExit("expected FactorizeError")
except G.FactorizeError:
    pass
dist.barrier()
dist.destroy_process_group()
open(os.path.join(%(out)r, "rank%%d.ok" %% rank), "w").write("ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_evaluation_matches_single_gpu(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                        "--master-port", str(29640 + world), str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in range(world):
        assert (tmp_path / ("rank%d.ok" % k)).exists()
