#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python tools/sharded_replay.py 3 2x4 16384 gpurun_out/r5_sharded_replay_c3.json > gpurun_out/r5_sharded_replay_c3.txt 2>&1 || { tail -20 gpurun_out/r5_sharded_replay_c3.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r5_sharded_replay_c3.txt
timeout -k 10 600 python tools/sharded_replay.py 4 2x4 32768 gpurun_out/r5_sharded_replay_c4.json > gpurun_out/r5_sharded_replay_c4.txt 2>&1 || { tail -20 gpurun_out/r5_sharded_replay_c4.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r5_sharded_replay_c4.txt
