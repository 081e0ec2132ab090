#!/bin/bash
# kernel traces of one evaluation with the chain forms side by side (N = 4096): eager = 0 and eager = 1
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
N=${1:-4096}
for cfg in "chain_split=2,eager=0" "chain_split=2,eager=1" "chain_split=1,eager=0"; do
  tag=$(echo $cfg | tr ',=' '__')
  export GOGP_OPTS=$cfg
  rocprofv3 --kernel-trace --output-format csv -d $O/trq_$tag -- python3 $R/tools/gpu_probe.py $N > $O/trq_$tag.log 2>&1
  python3 $R/tools/trace_queue.py $O/trq_$tag "" ${2:-45} > $O/trq_$tag.txt 2>&1
  grep steady $O/trq_$tag.log
  find $O/trq_$tag -name "*.db" -delete
done
head -70 $O/trq_chain_split_2_eager_0.txt
