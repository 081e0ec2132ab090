#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rf -k "chain_step or two_halves or candidates" > gpurun_out/r5v_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r5v_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/panel_probe.py 4096 2>&1 | grep -v amdgpu.ids | grep "rows_below" | tee gpurun_out/r5v_panel_probe.txt
timeout -k 10 300 python tools/split_probe.py 2048,4096,8192 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5v_split_probe.txt
