#!/bin/bash
# tools/exp/tune_env.sh CONFIG "VAR=val VAR=val" ... : bench runs under environment settings (probe builds only)
cfg=$1; shift
for grp in "$@"; do
  line=$(env $grp python3 bench.py --config $cfg --no-cpu-baseline --no-produce 2>/dev/null | tail -n 1)
  python3 - "$grp" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]
print("%-44s ms_per_step %.3f value %.3f frac %.4f" % (sys.argv[1], d["ms_per_step"], d["value"], r["frac"]))
PY
done
