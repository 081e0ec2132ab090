#!/bin/bash
# kernel trace of one evaluation at a tutorial-sized N: every launch in order, all queues
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trq_n64 -- python3 $R/tools/gpu_probe.py 64 > $O/trq_n64.log 2>&1
grep steady $O/trq_n64.log
python3 - <<PY
import csv, glob
f = glob.glob("$O/trq_n64/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void gogp::','').replace('gogp::','').split('(')[0][:46], r['Queue_Id']) for r in rows)
last = ev[-14:]
t0 = last[0][0]
for e in last:
    print("  %7.1f us  dur %6.1f  q%s  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3], e[2]))
PY
find $O/trq_n64 -name "*.db" -delete
