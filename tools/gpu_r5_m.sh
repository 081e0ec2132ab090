#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{ echo "== diag_fp64 = 0"; timeout -k 10 200 python tools/fp32_bias_probe.py 0; echo "== diag_fp64 = 1"; timeout -k 10 200 python tools/fp32_bias_probe.py 1; timeout -k 10 200 python tools/fp32_illcond_probe.py; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5m_fp32.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rf -k "fp32 or precision or float" > gpurun_out/r5m_tests.log 2>&1
rc=$?
tail -8 gpurun_out/r5m_tests.log
exit $rc
