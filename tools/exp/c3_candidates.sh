#!/bin/bash
# config 3 with k parameter vectors per launch sequence: tools/exp/c3_candidates.sh 2 3
for k in "$@"; do
  python3 bench.py --config 3 --candidates-per-step $k --steps 6 --warmup 2 --no-cpu-baseline --no-produce > gpurun_out/c3_k$k.json 2> gpurun_out/c3_k$k.err || tail -3 gpurun_out/c3_k$k.err
  python3 - $k <<'PY'
import json, sys
k = sys.argv[1]
d = json.loads(open("gpurun_out/c3_k%s.json" % k).read().strip().splitlines()[-1])
print(k, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_wall"))
PY
done
