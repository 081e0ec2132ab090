#!/bin/bash
# round 5, first GPU call: the one-pass Produce kernel (parity, then timing)
set -o pipefail
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -rf -k "few_points or superpanel_inverses or produce_known or smoke" > gpurun_out/r5a_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r5a_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/produce_small_probe.py 16384 8 > gpurun_out/r5a_probe.log 2>&1 || { tail -20 gpurun_out/r5a_probe.log; exit 1; }
cat gpurun_out/r5a_probe.log
timeout -k 10 200 python tools/produce_small_probe.py 4096 4 1,64 > gpurun_out/r5a_probe4k.log 2>&1; cat gpurun_out/r5a_probe4k.log
