#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
F="--no-cpu-baseline --no-produce --candidates 1"
for o in "" "--option superpanel=1" "--option superpanel=3" "--option head_remaining=8" "--option superpanel_head=2"; do
  echo "config 2 single $o"
  python bench.py --config 2 --candidates-per-step 1 --steps 40 --warmup 3 $F $o 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ms_per_step %.3f  value %.1f' % (d['ms_per_step'], d['value']))"
done
timeout -k 10 300 python tools/mixed_probe.py 16384 8 2>&1 | grep -v amdgpu | tail -12
