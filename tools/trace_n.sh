# usage: bash tools/trace_n.sh N   -- kernel trace of tools/gpu_probe.py N into gpurun_out/trace_n$N
set -e
N=${1:-4096}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_n$N -- python3 $R/tools/gpu_probe.py $N > $O/trace_n$N.log 2>&1
