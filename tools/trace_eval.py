"""One evaluation out of a rocprofv3 kernel trace (CSV): per hardware queue its launches, busy time and span, and the
chain queue (the one with the diagonal-block kernel) launch by launch -- start, gap to the previous launch on that
queue, duration, workgroups -- for a window of super-panels.
usage: python3 tools/trace_eval.py TRACE.csv [evaluation index, default 3] [first diag, default 18] [count, default 7]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d0 = int(sys.argv[3]) if len(sys.argv) > 3 else 18
dn = int(sys.argv[4]) if len(sys.argv) > 4 else 7
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['n'] = r['Kernel_Name'].replace('void gogp::', '').split('(')[0][:40]; r['q'] = r['Queue_Id']
    r['g'] = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
grams = sorted(r['s'] for r in rows if r['n'].startswith('gram_kernel'))
starts = [grams[0]] + [b for a, b in zip(grams, grams[1:]) if b - a > 5_000_000]
t0 = starts[ev]; t1 = starts[ev + 1] if ev + 1 < len(starts) else max(r['e'] for r in rows) + 1
E = sorted([r for r in rows if t0 - 20000 <= r['s'] < t1 - 20000], key=lambda r: r['s'])
print("evaluation %d: span %.3f ms, %d launches" % (ev, (max(r['e'] for r in E) - t0) / 1e6, len(E)))
byq = collections.defaultdict(list)
for r in E:
    byq[r['q']].append(r)
for q, l in sorted(byq.items()):
    busy = sum(r['e'] - r['s'] for r in l) / 1e6
    print("queue %s: %3d launches, busy %.2f ms, first start %.3f, last end %.3f  %s" % (
        q, len(l), busy, (l[0]['s'] - t0) / 1e6, (l[-1]['e'] - t0) / 1e6, collections.Counter(r['n'] for r in l).most_common(4)))
cq = [q for q, l in byq.items() if any(r['n'].startswith('diag256') for r in l)][0]
l = byq[cq]
di = [i for i, r in enumerate(l) if r['n'].startswith('diag256')]
print("chain queue %s, from diagonal block %d:" % (cq, d0))
for i in range(di[d0], di[min(d0 + dn, len(di) - 1)]):
    r = l[i]; prev = l[i - 1] if i else r
    print("%8.3f ms  gap %7.1f us  dur %7.1f us  wgs %5d  %s" % ((r['s'] - t0) / 1e6, (r['s'] - prev['e']) / 1e3, (r['e'] - r['s']) / 1e3, r['g'], r['n']))
