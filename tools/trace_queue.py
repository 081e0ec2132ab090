"""Timeline of ONE queue inside the last evaluation of a rocprofv3 kernel trace: the queue that runs the kernel whose name
contains `key` (default: the Cholesky chain's panel128 / diag256 / diag128 kernels), launch by launch: start, duration,
gap to the previous launch on that queue; then the totals per kernel name.
usage: python3 tools/trace_queue.py <trace dir> [key] [max lines]"""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
key = sys.argv[2] if len(sys.argv) > 2 else None
maxl = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = list(csv.DictReader(open(f)))
def short(n):
    n = n.replace('void gogp::', '').replace('gogp::', '')
    return n.split('(')[0][:40]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?'),
             int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0), int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1))
            for r in rows)
grams = [i for i, e in enumerate(ev) if 'gram_kernel' in e[2]]
last = ev[grams[-2]:] if len(grams) >= 2 else ev
t0 = last[0][0]
tend = max(e[1] for e in last)
print("evaluation span %.3f ms, %d kernels" % ((tend - t0) / 1e6, len(last)))
keys = [key] if key else ['panel128', 'diag128', 'diag256']
q = None
for k in keys:
    hit = [e for e in last if k in e[2]]
    if hit:
        q = hit[0][3]
        break
chain = [e for e in last if e[3] == q]
print("queue %s: %d launches, first start %.3f ms, last end %.3f ms" % (q, len(chain), (chain[0][0] - t0) / 1e6, (chain[-1][1] - t0) / 1e6))
prev = chain[0][0]
agg = {}
for i, e in enumerate(chain):
    if i < maxl:
        print("  %8.1f us  dur %7.1f  gap %6.1f  wgs %5d  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, (e[0] - prev) / 1e3, e[4] // max(1, e[5]), e[2]))
    a = agg.setdefault(e[2], [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e[1] - e[0]) / 1e3
    a[2] += max(0.0, (e[0] - prev) / 1e3)
    prev = e[1]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  total %-40s x%4d  busy %9.1f us (avg %7.1f)  gaps before %8.1f us" % (k, v[0], v[1], v[1] / v[0], v[2]))
