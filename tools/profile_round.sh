set -e
# the JSON line comes from the default command; the rocprofv3 passes skip the CPU baseline and the
# secondary Produce measurement so that their per-kernel sums are per Observe+Gradient evaluation
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/r1_bench.json 2> $O/r1_bench.err
rm -rf $O/r1_prof $O/r1_pmc_sq $O/r1_pmc_fetch $O/r1_pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r1_prof -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-produce > $O/r1_prof.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/r1_pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-produce > $O/r1_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r1_pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-produce > $O/r1_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/r1_pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-produce > $O/r1_pmc_write.log 2>&1
cd $R && python3 tools/pmc_summary.py gpurun_out/r1_pmc_sq gpurun_out/r1_pmc_fetch gpurun_out/r1_pmc_write > $O/r1_pmc_summary.txt
find $O/r1_pmc_sq $O/r1_pmc_fetch $O/r1_pmc_write -name "*counter_collection.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -size +8M -delete
