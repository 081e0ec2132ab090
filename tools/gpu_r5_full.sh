#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -rf > gpurun_out/r5_full_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r5_full_tests.log
exit $rc
