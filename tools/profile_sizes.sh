# rocprofv3 kernel-stat summaries of the probe runs behind DESIGN.md's size table (N = 4096 and N = 32768)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r1_prof_n4096 $O/r1_prof_n32768
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r1_prof_n4096 -- python3 $R/tools/gpu_probe.py 4096 > $O/r1_prof_n4096.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r1_prof_n32768 -- python3 $R/tools/big_probe.py 32768 > $O/r1_prof_n32768.log 2>&1
find $O/r1_prof_n4096 $O/r1_prof_n32768 -name "*kernel_trace.csv" -delete
