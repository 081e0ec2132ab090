#!/bin/bash
# per-workgroup time stamps of one evaluation with the probe build (tools/wg_stamps.py); restores the product library
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python tools/trsv_stamps.py 16384 > gpurun_out/r5_trsv_stamps.txt 2>&1; tail -9 gpurun_out/r5_trsv_stamps.txt
cp gogp_amd/libgogp_hip.so /tmp/product.so
cp tools/exp/lib_stamp.so gogp_amd/libgogp_hip.so
for cfg in "16384 8 1" "16384 8 0" "4096 4 1"; do
  set -- $cfg
  timeout -k 10 300 python tools/wg_stamps.py $1 $2 $3 gpurun_out/r5_wg_stamps_n$1_e$3.json > gpurun_out/r5_wg_stamps_n$1_e$3.txt 2>&1 || { tail -20 gpurun_out/r5_wg_stamps_n$1_e$3.txt; cp /tmp/product.so gogp_amd/libgogp_hip.so; exit 1; }
  cat gpurun_out/r5_wg_stamps_n$1_e$3.txt | cut -c1-330
done
cp /tmp/product.so gogp_amd/libgogp_hip.so
