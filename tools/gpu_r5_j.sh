#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -rf > gpurun_out/r5j_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r5j_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 120 python tools/diag_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/r5j_diag_probe.txt
timeout -k 10 300 python tools/split_probe.py 4096,16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5j_split_probe.txt
