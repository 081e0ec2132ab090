#!/bin/bash
cd $GRAFT_REPO_ROOT
cp gogp_amd/libgogp_hip.so /tmp/p.so
for e in 0 1 2 3; do
  if [ $e = 0 ]; then cp /tmp/p.so gogp_amd/libgogp_hip.so; else cp tools/exp/lib_ts$e.so gogp_amd/libgogp_hip.so; fi
  echo "variant $e (bit 0: no FMAs in the L phase, bit 1: no loads of L in the loop)"
  timeout -k 10 120 python tools/produce_small_probe.py 16384 8 1 2>&1 | grep "one-pass"
  timeout -k 10 120 python tools/produce_small_probe.py 4096 4 1 2>&1 | grep "one-pass"
done
cp /tmp/p.so gogp_amd/libgogp_hip.so
