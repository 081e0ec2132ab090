set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 -m pytest tests -m gpu -x -q > $O/r2c_tests.log 2>&1 || { tail -40 $O/r2c_tests.log; exit 1; }
tail -3 $O/r2c_tests.log
export GOGP_DIST_BACKEND=gloo
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 4 --config 4 --nobs 8192 --steps 2 --warmup 1 > $O/r2c_bench_c4_g4.json 2> $O/r2c_bench_c4_g4.err || { tail -30 $O/r2c_bench_c4_g4.err; exit 1; }
cat $O/r2c_bench_c4_g4.json
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 2 --config 3 --nobs 8192 --steps 2 --warmup 1 > $O/r2c_bench_c3_g2.json 2> $O/r2c_bench_c3_g2.err || { tail -30 $O/r2c_bench_c3_g2.err; exit 1; }
cat $O/r2c_bench_c3_g2.json
unset GOGP_DIST_BACKEND
python3 - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
from gogp_amd import configs, kernel
from gogp_amd.sharded import ShardedGP
from gogp_amd import gp as G
for cfg in (3, 4):
    wl = configs.workload(cfg)
    X, y = wl.inputs()
    sh = ShardedGP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0, transport="rccl")
    g = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0)
    for obj, name in ((sh, "sharded 1x1 (2-D code path, RCCL transport)"), (g, "single-GPU fused sweep")):
        obj.Observe(wl.log_theta(0)); obj.Gradient()
        t0 = time.perf_counter()
        for k in range(3):
            lml = obj.Observe(wl.log_theta(1 + k)); gr = obj.Gradient()
        dt = (time.perf_counter() - t0) / 3
        print("config %d %s: %.1f ms/eval lml=%.9f" % (cfg, name, dt * 1e3, lml), flush=True)
    sh.close(); g.close()
PY
