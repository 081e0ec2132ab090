#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_sharded.py -x -q -m gpu -rf -k "fp32 or float or diagonal_tiles" > gpurun_out/r5t_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r5t_tests.log
exit $rc
