set -e
# Round-5 profiles in ONE call on one box (profiles/README.md, round 5): HBM traffic per launch first (separate --pmc passes at
# the configurations' own sizes; written into profiles/ of the box's copy so that the bench lines that follow carry
# `roofline.traffic` of this very build), then the bench lines of all five configurations, the rocprofv3 kernel-trace
# summaries, the SQ counter passes, the probes.  PMC passes have only --kernel-trace beside --pmc.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r5
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAST="--no-cpu-baseline --no-produce --candidates 1 --no-sharded"
for c in 2 3 4 5; do
  st=2; [ $c = 2 ] && st=5
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c$c -- python3 $R/bench.py --config $c --steps $st --warmup 1 $FAST > $O/pmc_fetch_c$c.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write_c$c -- python3 $R/bench.py --config $c --steps $st --warmup 1 $FAST > $O/pmc_write_c$c.log 2>&1
  echo "pmc traffic c$c done"
done
cd $R
python3 tools/pmc_traffic.py 2 $O/pmc_fetch_c2 $O/pmc_write_c2 dgemm_nt_kernel 4096 > $O/t2.json
python3 tools/pmc_traffic.py 3 $O/pmc_fetch_c3 $O/pmc_write_c3 dgemm_nt_kernel 16384 > $O/t3.json
python3 tools/pmc_traffic.py 4 $O/pmc_fetch_c4 $O/pmc_write_c4 dgemm_nt_kernel 32768 > $O/t4.json
python3 tools/pmc_traffic.py 5 $O/pmc_fetch_c5 $O/pmc_write_c5 sgemm_nt_kernel 65536 > $O/t5.json
python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
from gogp_amd import _lib
v = _lib.lib().gogp_version().decode()
d = {"build": v.split("build ")[-1], "library": v}
for n in ("t2", "t3", "t4", "t5"):
    d.update(json.load(open("$O/%s.json" % n)))
d["2"]["candidates_per_step"] = 8
d["2"]["note"] += "; config 2 as its line runs it: 8 candidates per launch sequence, so a launch carries 8 tiles' worth"
json.dump(d, open("$O/pmc_traffic.json", "w"), indent=1)
json.dump(d, open("$R/profiles/r05_pmc_traffic.json", "w"), indent=1)
print(v)
PY
cd /tmp
python3 $R/bench.py --sharded > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 done"
for c in 1 2 4 5; do python3 $R/bench.py --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err; echo "bench c$c done"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py --steps 5 --warmup 1 $FAST > $O/stats_c3.log 2>&1; echo "stats c3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 $R/bench.py --config 2 --candidates-per-step 1 --steps 20 --warmup 2 $FAST > $O/stats_c2.log 2>&1; echo "stats c2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2k8 -- python3 $R/bench.py --config 2 --steps 10 --warmup 2 $FAST > $O/stats_c2k8.log 2>&1; echo "stats c2k8 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py --config 5 --nobs 32768 --steps 2 --warmup 1 $FAST > $O/stats_c5.log 2>&1; echo "stats c5 done"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_sq_c3.log 2>&1; echo "pmc sq c3 done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq_c5 -- python3 $R/bench.py --config 5 --steps 1 --warmup 1 $FAST > $O/pmc_sq_c5.log 2>&1; echo "pmc sq c5 (N = 65536) done"
cd $R
python3 tools/pmc_summary.py $O/pmc_sq_c3 $O/pmc_fetch_c3 $O/pmc_write_c3 > $O/pmc_summary_c3.txt
python3 tools/pmc_summary.py $O/pmc_sq_c5 $O/pmc_fetch_c5 $O/pmc_write_c5 > $O/pmc_summary_c5.txt
python3 tools/roofline_from_profiles.py 16384 dgemm_nt_kernel 78.6 $O/stats_c3 $O/pmc_sq_c3 > $O/roofline_c3.json
python3 tools/roofline_from_profiles.py 4096 dgemm_nt_kernel 78.6 $O/stats_c2 > $O/roofline_c2.json
python3 tools/roofline_from_profiles.py 32768 sgemm_nt_kernel 157.3 $O/stats_c5 > $O/roofline_c5_n32768.json
python3 tools/produce_probe.py 16384 8 "produce_small_max=0" 256,1024,4096 2>&1 | grep -v amdgpu.ids > $O/produce_probe.txt; echo "produce probe done"
python3 tools/produce_small_probe.py 16384 8 1,2,8,16,32,64 2>&1 | grep -v amdgpu.ids > $O/produce_small_probe.txt; echo "small produce probe done"
python3 tools/mixed_probe.py 16384 8 2>&1 | grep -v amdgpu.ids > $O/mixed_probe.txt; echo "mixed probe done"
python3 tools/split_probe.py 1024,2048,4096,8192,16384 2>&1 | grep -v amdgpu.ids > $O/split_probe.txt; echo "split probe done"
python3 tools/panel_probe.py 0,4096,16256 2>&1 | grep -v amdgpu.ids > $O/panel_probe.txt
python3 tools/valu_cost.py 2>&1 | grep -v amdgpu.ids > $O/valu_cost.txt
{ echo "== diag_fp64 = 0"; python3 tools/fp32_bias_probe.py 0; echo "== diag_fp64 = 1"; python3 tools/fp32_bias_probe.py 1; python3 tools/fp32_illcond_probe.py; } 2>&1 | grep -v amdgpu.ids > $O/fp32_bias_probe.txt; echo "kernel probes done"
python3 tools/graph_probe.py 64,1024,4096 1,8 2>&1 | grep -v amdgpu.ids > $O/graph_probe.txt; echo "graph probe done"
find $O -name "*counter_collection.csv" -size +6M -delete
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*.db" -delete
tail -c 300 $O/bench_c3.json
