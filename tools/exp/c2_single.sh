#!/bin/bash
# config 2, one evaluation at a time (what a sequential optimiser sees) and 8 per launch sequence: prints both
for i in 1 2; do
python3 bench.py --config 2 --no-cpu-baseline --no-produce > gpurun_out/c2_single_$i.json 2> gpurun_out/c2_single_$i.err || tail -3 gpurun_out/c2_single_$i.err
python3 - $i <<'PY'
import json, sys
d = json.loads(open("gpurun_out/c2_single_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("k=8: %.1f evals/s (%.3f ms per step); single: %s" % (d["value"], d["ms_per_step"], {k: d["single_candidate"][k] for k in ("evals_per_s", "ms_per_eval")}))
PY
done
for c in 1; do python3 bench.py --config 1 --no-cpu-baseline --no-produce | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 1:', d['value'], d['ms_per_step'], d.get('single_candidate',{}).get('ms_per_eval'))"; done
