#!/bin/bash
# the round's last call: the whole GPU suite, then every profile of profiles/r05_* on the same build
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -rf > gpurun_out/r5_final_tests.log 2>&1
rc=$?
tail -6 gpurun_out/r5_final_tests.log
[ $rc = 0 ] || exit $rc
bash tools/profile_round5.sh > gpurun_out/prof_r5.log 2>&1
rc=$?
tail -4 gpurun_out/prof_r5.log
exit $rc
