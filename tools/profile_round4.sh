set -e
# Round-4 profiles (one call on one box).  Per profiled configuration: the bench line, the
# `rocprofv3 --kernel-trace --stats` per-kernel summary, the kernel-trace-derived union / sum of the
# dominant kernel's launch intervals and the serialised PMC time (tools/roofline_from_profiles.py), the
# per-family PMC sums, and the HBM traffic per launch stamped with the library's build id.
# PMC passes are separate runs with only --kernel-trace beside --pmc.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAST="--no-cpu-baseline --no-produce --candidates 1"
python3 $R/bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 done"
for c in 1 2 4 5; do python3 $R/bench.py --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err; echo "bench c$c done"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py --steps 5 --warmup 1 $FAST > $O/stats_c3.log 2>&1; echo "stats c3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 $R/bench.py --config 2 --candidates-per-step 1 --steps 20 --warmup 2 $FAST > $O/stats_c2.log 2>&1; echo "stats c2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2k8 -- python3 $R/bench.py --config 2 --steps 10 --warmup 2 $FAST > $O/stats_c2k8.log 2>&1; echo "stats c2k8 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py --config 5 --nobs 32768 --steps 2 --warmup 1 $FAST > $O/stats_c5.log 2>&1; echo "stats c5 done"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_sq_c3.log 2>&1; echo "pmc sq c3 done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_fetch_c3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_write_c3.log 2>&1; echo "pmc traffic c3 done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq_c5 -- python3 $R/bench.py --config 5 --nobs 16384 --steps 2 --warmup 1 $FAST > $O/pmc_sq_c5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c5 -- python3 $R/bench.py --config 5 --nobs 16384 --steps 2 --warmup 1 $FAST > $O/pmc_fetch_c5.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write_c5 -- python3 $R/bench.py --config 5 --nobs 16384 --steps 2 --warmup 1 $FAST > $O/pmc_write_c5.log 2>&1; echo "pmc c5 done"
cd $R
python3 tools/gemm_fixedcost.py 2>&1 | grep -v amdgpu.ids > $O/gemm_fixedcost.txt; echo "fixed cost done"
python3 tools/graph_probe.py 64,512,1024,2048,4096,8192 1,8 2>&1 | grep -v amdgpu.ids > $O/graph_probe.txt; echo "graph probe done"
python3 tools/produce_probe.py 16384 8 "" 1,64,256,1024,4096 2>&1 | grep -v amdgpu.ids > $O/produce_probe.txt; echo "produce probe done"
python3 tools/mixed_probe.py 16384 8 2>&1 | grep -v amdgpu.ids > $O/mixed_probe.txt; echo "mixed probe done"
{ python3 tools/gemm_bench.py; GOGP_BENCH_GEMM_LD0=1 python3 tools/gemm_bench.py; GOGP_BENCH_GEMM_F32=1 python3 tools/gemm_bench.py; } 2>&1 | grep -v amdgpu.ids > $O/gemm_bench.txt; echo "gemm bench done"
python3 tools/pmc_summary.py $O/pmc_sq_c3 $O/pmc_fetch_c3 $O/pmc_write_c3 > $O/pmc_summary_c3.txt
python3 tools/pmc_summary.py $O/pmc_sq_c5 $O/pmc_fetch_c5 $O/pmc_write_c5 > $O/pmc_summary_c5.txt
python3 tools/roofline_from_profiles.py 16384 dgemm_nt_kernel 78.6 $O/stats_c3 $O/pmc_sq_c3 > $O/roofline_c3.json
python3 tools/roofline_from_profiles.py 4096 dgemm_nt_kernel 78.6 $O/stats_c2 > $O/roofline_c2.json
python3 tools/roofline_from_profiles.py 32768 sgemm_nt_kernel 157.3 $O/stats_c5 > $O/roofline_c5_n32768.json
python3 tools/pmc_traffic.py 3 $O/pmc_fetch_c3 $O/pmc_write_c3 dgemm_nt_kernel 16384 > $O/t3.json
python3 tools/pmc_traffic.py 5_at_N16384 $O/pmc_fetch_c5 $O/pmc_write_c5 sgemm_nt_kernel 16384 > $O/t5.json
python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
from gogp_amd import _lib
v = _lib.lib().gogp_version().decode()
d = {"build": v.split("build ")[-1], "library": v}
d.update(json.load(open("$O/t3.json"))); d.update(json.load(open("$O/t5.json")))
json.dump(d, open("$O/pmc_traffic.json", "w"), indent=1)
print(v)
PY
find $O -name "*counter_collection.csv" -size +6M -delete
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*.db" -delete
ls -la $O
