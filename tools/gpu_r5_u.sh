#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu -rf > gpurun_out/r5u_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r5u_tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 260 python tools/stress_sharded.py 180 21 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r5u_stress_sharded.txt
timeout -k 10 260 python tools/stress.py 180 31 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r5u_stress.txt
