#!/bin/bash
# tools/exp/ab_lib.sh CONFIG N : alternate the built library and tools/exp/lib_base.so, N pairs
cfg=$1; n=$2
cp gogp_amd/libgogp_hip.so /tmp/new.so
for i in $(seq 1 $n); do
  for which in new base; do
    if [ $which = new ]; then cp /tmp/new.so gogp_amd/libgogp_hip.so; else cp tools/exp/lib_base.so gogp_amd/libgogp_hip.so; fi
    python3 bench.py --config $cfg --no-cpu-baseline --no-produce 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$which', round(d['ms_per_step'],3), round(d['value'],3), round(d['roofline']['frac'],4))"
  done
done
cp /tmp/new.so gogp_amd/libgogp_hip.so
