#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rf -k "fp32 or produce or Produce or precision" > gpurun_out/r5y_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r5y_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5y_fp32_small_produce.txt
import sys, time
sys.path.insert(0, ".")
import numpy as np
from gogp_amd import configs, gp as G
for n in (16384, 65536):
    wl = configs.workload(5, n)
    X, y = wl.inputs()
    g = G.GP(wl.D, wl.simil, wl.noise, X=X, Y=y, precision=32)
    g.Observe(wl.log_theta(0)); g.Gradient()
    for m in (1, 16, 64):
        Z = wl.test_points(m)
        out = {}
        for small in (64, 0):
            g.set_option("produce_small_max", small)
            g.Produce(Z)
            t = time.perf_counter()
            for _ in range(3):
                mu, sg = g.Produce(Z)
            out[small] = ((time.perf_counter() - t) / 3 * 1e3, mu, sg)
        print("fp32 N %d M %2d: one pass %.3f ms, tile-kernel chain %.3f ms; mu diff %.2e sigma diff %.2e" %
              (n, m, out[64][0], out[0][0], np.abs(out[64][1] - out[0][1]).max() / np.abs(out[0][1]).max(),
               np.abs(out[64][2] - out[0][2]).max() / np.abs(out[0][2]).max()), flush=True)
    g.close()
PY
