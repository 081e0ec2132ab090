#!/bin/bash
# kernel trace of Observe only (eager = 0) at N = 16384 with the one-launch chain step: the chain queue and the bulk queue
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export GOGP_OPTS="chain_split=2,eager=0"
rocprofv3 --kernel-trace --output-format csv -d $O/trq_16k -- python3 $R/tools/gpu_probe.py 16384 > $O/trq_16k.log 2>&1
grep steady $O/trq_16k.log
python3 $R/tools/trace_queue.py $O/trq_16k "" 0 | tail -12
python3 $R/tools/trace_queue.py $O/trq_16k "dgemm_nt_kernel<1" 0 | tail -8
find $O/trq_16k -name "*.db" -delete
