"""Timeline of the panel chain inside the last evaluation of a kernel trace: for every diagonal-block
kernel its start, duration, and what ran between the end of the previous one and its start."""
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=__import__("os").path.getmtime)
rows = list(csv.DictReader(open(f)))
def short(n):
    n = n.replace('void gogp::', '').replace('gogp::', '')
    return n.split('(')[0][:34]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?')) for r in rows)
grams = [i for i, e in enumerate(ev) if 'gram_kernel' in e[2]]
# evaluations start with 2 gram launches (split build): take the last pair
last = ev[grams[-2]:] if len(grams) >= 2 else ev
t0 = last[0][0]
tend = max(e[1] for e in last)
print("eval span %.3f ms, %d kernels" % ((tend - t0) / 1e6, len(last)))
diags = [e for e in last if 'diag256' in e[2]]
prev_end = t0
for i, d in enumerate(diags):
    between = [e for e in last if e[1] > prev_end and e[0] < d[0] and e is not d and e[3] == d[3]]
    names = ", ".join("%s %.0fus" % (e[2][:22], (e[1] - e[0]) / 1e3) for e in between[:6])
    print("diag %2d: start %7.3f ms dur %6.1f us  gap since prev diag end %6.1f us | same-queue: %s" % (
        i, (d[0] - t0) / 1e6, (d[1] - d[0]) / 1e3, (d[0] - prev_end) / 1e3, names))
    prev_end = d[1]
print("after last diag: %.3f ms to the end" % ((tend - prev_end) / 1e6))
tail = [e for e in last if e[0] >= prev_end]
agg = {}
for e in tail:
    agg.setdefault(e[2], [0, 0.0])
    agg[e[2]][0] += 1
    agg[e[2]][1] += (e[1] - e[0]) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print("   tail: %-36s x%3d  %8.1f us" % (k, v[0], v[1]))
