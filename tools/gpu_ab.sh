# A/B: libgogp_hip.so (new) against libgogp_hip_base.so, config 3 and config 2 one at a time
cp gogp_amd/libgogp_hip.so /tmp/new.so
for i in 1 2; do
for v in new base; do
if [ $v = base ]; then cp gogp_amd/libgogp_hip_base.so gogp_amd/libgogp_hip.so; else cp /tmp/new.so gogp_amd/libgogp_hip.so; fi
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-produce --candidates 1 --steps 10 > gpurun_out/ab_${v}_c3_$i.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --config 2 --candidates-per-step 1 --no-cpu-baseline --no-produce --candidates 1 --steps 40 > gpurun_out/ab_${v}_c2_$i.json 2>/dev/null || exit 1
done
done
cp /tmp/new.so gogp_amd/libgogp_hip.so
python3 - <<'PY'
import json
for c in ("c3", "c2"):
    for v in ("new", "base"):
        for i in (1, 2):
            d = json.load(open("gpurun_out/ab_%s_%s_%d.json" % (v, c, i))); print(c, v, i, round(d["value"], 3), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4))
PY
