#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rf -k "few_points or superpanel_inverses or produce_known" > gpurun_out/r5b_tests.log 2>&1 || { tail -30 gpurun_out/r5b_tests.log; exit 1; }
tail -3 gpurun_out/r5b_tests.log
timeout -k 10 200 python tools/trsv_stamps.py 16384 > gpurun_out/r5_trsv_stamps.txt 2>&1; tail -8 gpurun_out/r5_trsv_stamps.txt
timeout -k 10 300 python tools/produce_small_probe.py 16384 8 > gpurun_out/r5b_probe.log 2>&1 || { tail -20 gpurun_out/r5b_probe.log; exit 1; }
grep -v "amdgpu.ids" gpurun_out/r5b_probe.log
