set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_srv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-produce > $O/trace_srv.log 2>&1
