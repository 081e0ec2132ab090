set -e
# Round-5 profiles, part A (one call on one box): the bench lines of all five configurations, the rocprofv3 kernel-trace
# summaries, the PMC passes of config 3.  PMC passes are separate runs with only --kernel-trace beside --pmc.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
FAST="--no-cpu-baseline --no-produce --candidates 1 --no-sharded"
python3 $R/bench.py --sharded > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 done"
for c in 1 2 4 5; do python3 $R/bench.py --config $c > $O/bench_c$c.json 2> $O/bench_c$c.err; echo "bench c$c done"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py --steps 5 --warmup 1 $FAST > $O/stats_c3.log 2>&1; echo "stats c3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -- python3 $R/bench.py --config 2 --candidates-per-step 1 --steps 20 --warmup 2 $FAST > $O/stats_c2.log 2>&1; echo "stats c2 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2k8 -- python3 $R/bench.py --config 2 --steps 10 --warmup 2 $FAST > $O/stats_c2k8.log 2>&1; echo "stats c2k8 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py --config 5 --nobs 32768 --steps 2 --warmup 1 $FAST > $O/stats_c5.log 2>&1; echo "stats c5 done"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_sq_c3.log 2>&1; echo "pmc sq c3 done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_fetch_c3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_write_c3 -- python3 $R/bench.py --steps 2 --warmup 1 $FAST > $O/pmc_write_c3.log 2>&1; echo "pmc traffic c3 done"
cd $R
python3 tools/pmc_summary.py $O/pmc_sq_c3 $O/pmc_fetch_c3 $O/pmc_write_c3 > $O/pmc_summary_c3.txt
python3 tools/roofline_from_profiles.py 16384 dgemm_nt_kernel 78.6 $O/stats_c3 $O/pmc_sq_c3 > $O/roofline_c3.json
python3 tools/roofline_from_profiles.py 4096 dgemm_nt_kernel 78.6 $O/stats_c2 > $O/roofline_c2.json
python3 tools/roofline_from_profiles.py 32768 sgemm_nt_kernel 157.3 $O/stats_c5 > $O/roofline_c5_n32768.json
python3 tools/pmc_traffic.py 3 $O/pmc_fetch_c3 $O/pmc_write_c3 dgemm_nt_kernel 16384 > $O/t3.json
find $O -name "*counter_collection.csv" -size +6M -delete
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*.db" -delete
ls $O
