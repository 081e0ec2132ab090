"""Largest gaps (no dgemm kernel running) inside the last evaluation of a kernel trace."""
import csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=__import__("os").path.getmtime)
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
grams = [i for i, e in enumerate(ev) if 'gram_kernel' in e[2]]
last = ev[grams[-1]:]
if len(grams) > 1:
    prev_end = max(e[1] for e in ev[grams[-2]:grams[-1]])
    print("gap between evaluations (prev last kernel end -> this gram start): %.3f ms" % ((last[0][0] - prev_end) / 1e6))
t0 = last[0][0]
tend = max(e[1] for e in last)
print("eval span %.2f ms" % ((tend - t0) / 1e6))
gem = sorted((s, e) for s, e, n in last if 'dgemm' in n)
gaps = []
cur = t0
for s, e in gem:
    if s > cur:
        gaps.append((s - cur, cur - t0))
    cur = max(cur, e)
if tend > cur:
    gaps.append((tend - cur, cur - t0))
tot = sum(g for g, _ in gaps)
print("total no-gemm time %.2f ms in %d gaps" % (tot / 1e6, len(gaps)))
for g, at in sorted(gaps, reverse=True)[:12]:
    names = sorted({n.split('(')[0].replace('void gogp::', '').replace('gogp::', '')[:28] for s, e, n in last
                    if s < t0 + at + g and e > t0 + at and 'dgemm' not in n})
    print("  gap %7.1f us at t=%7.2f ms : %s" % (g / 1e3, at / 1e6, ", ".join(names)))
