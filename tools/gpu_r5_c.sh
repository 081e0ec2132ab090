#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rf -k "two_halves or not_positive or known_answers or small_vs_faithful or ragged or candidates" > gpurun_out/r5c_tests.log 2>&1 || { tail -40 gpurun_out/r5c_tests.log; exit 1; }
tail -3 gpurun_out/r5c_tests.log
timeout -k 10 600 python tools/split_probe.py > gpurun_out/r5c_split.log 2>&1 || { tail -20 gpurun_out/r5c_split.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r5c_split.log
