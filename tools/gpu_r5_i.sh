#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rf -k "chain_step or two_halves" > gpurun_out/r5i_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r5i_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 300 python tools/panel_probe.py ${1:-0,1024,4096,16256} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5i_panel_probe.txt
