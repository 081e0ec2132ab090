#!/bin/bash
# final build: the per-rank replays again (the diagonal-block kernel changed since the round's first build) and long stress runs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python tools/sharded_replay.py 4 2x4 32768 gpurun_out/r5x_replay_c4.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5x_replay_c4.txt
timeout -k 10 200 python tools/sharded_replay.py 3 2x4 16384 gpurun_out/r5x_replay_c3.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5x_replay_c3.txt
timeout -k 10 460 python tools/stress.py 420 41 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r5x_stress.txt
timeout -k 10 340 python tools/stress_sharded.py 300 42 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r5x_stress_sharded.txt
