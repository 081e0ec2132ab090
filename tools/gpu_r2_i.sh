R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_c2
rocprofv3 --kernel-trace --output-format csv -d $O/trace_c2 -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --no-produce --candidates 1 > $O/trace_c2.log 2>&1
cd $R && python3 tools/trace_chain.py gpurun_out/trace_c2 > $O/trace_c2_chain.txt; cat $O/trace_c2_chain.txt
find $O/trace_c2 -name "*.csv" -size +8M -delete
