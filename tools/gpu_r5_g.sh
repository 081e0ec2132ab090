#!/bin/bash
# chain_split = 2 (panel128.hip): its parity test first, then the A/B timings of the three chain forms
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rf -k "two_halves" > gpurun_out/r5g_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r5g_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 500 python tools/split_probe.py ${1:-1024,4096,8192,16384} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5g_split_probe.txt
