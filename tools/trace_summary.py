"""Summarise a rocprofv3 kernel-trace CSV: last evaluation's per-kernel busy time per queue,
panel periods, diag/SYRK durations."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
f = max(glob.glob(path + "/*/*_kernel_trace.csv"), key=__import__("os").path.getmtime)
rows = list(csv.DictReader(open(f)))
KEYS = ['dgemm_nt_kernel<0, 64>', 'dgemm_nt_kernel<1, 64>', 'dgemm_nt_kernel<0, 128>', 'dgemm_nt_kernel<1, 128>',
        'dgemm_nt_kernel<2, 128>', 'diag256', 'diag128', 'trsv_fwd', 'trsv_bwd', 'grad_reduce', 'grad_final',
        'gram_kernel', 'identity_upper', 'lml_scalars', 'mfma_f64_peak', 'copyBuffer', 'fillBuffer']


def short(n):
    for k in KEYS:
        if k in n:
            return k
    return n[:40]


ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), int(r['Queue_Id']))
            for r in rows)
grams = [i for i, e in enumerate(ev) if e[2] == 'gram_kernel']
last = ev[grams[-1]:]
t0 = last[0][0]
tend = max(e[1] for e in last)
print("last eval span ms: %.2f" % ((tend - t0) / 1e6))
idn = [e for e in last if e[2] == 'identity_upper']
if idn:
    print("observe phase ms: %.2f  gradient phase ms: %.2f" % ((idn[0][0] - t0) / 1e6, (tend - idn[0][0]) / 1e6))
byq = collections.defaultdict(lambda: [0.0, 0])
for e in last:
    byq[(e[3], e[2])][0] += (e[1] - e[0]) / 1e6
    byq[(e[3], e[2])][1] += 1
for k, v in sorted(byq.items()):
    print("  queue %d %-26s busy %8.2f ms  calls %4d  avg %8.1f us" % (k[0], k[1], v[0], v[1], v[0] / v[1] * 1e3))
for name in ('diag256', 'diag128'):
    d = [e for e in last if e[2] == name]
    if len(d) > 2:
        per = [(d[i + 1][0] - d[i][0]) / 1e3 for i in range(len(d) - 1)]
        print(name, "dur us every 8th:", [round((e[1] - e[0]) / 1e3) for e in d[::8]])
        print(name, "period us every 8th:", [round(x) for x in per[::8]])
lows = [e for e in last if e[2].startswith('dgemm_nt_kernel<1')]
print("syrk dur us every 8th:", [round((e[1] - e[0]) / 1e3) for e in lows[::8]])
