"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel family."""
import collections, csv, glob, sys

import re


def short(n):
    m = re.search(r'dgemm_nt_kernel<(\d), (\d+), (\d)>', n)
    if m:  # MODE, TILE, WAVES
        return 'dgemm_nt_kernel<%s, %s, %s>' % m.group(1, 2, 3)
    m = re.search(r'sgemm_nt_kernel<(\d), (\d+), (\d)>', n)
    if m:  # MODE, TILE, WAVES
        return 'sgemm_nt_kernel<%s, %s, %s>' % m.group(1, 2, 3)
    for k in ['diag256', 'diag128', 'trsv_granule', 'trsm_small', 'blockmm', 'trsv_fwd', 'trsv_bwd', 'grad_reduce', 'gram_kernel', 'alpha_from_y', 'zero_upper', 'kmatvec', 'convert_block',
              'mfma_f64_peak', 'mfma_f32_peak']:
        if k in n:
            return k
    return None


for path in sys.argv[1:]:
    f = max(glob.glob(path + "/*/*_counter_collection.csv"), key=__import__("os").path.getmtime)
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = {}
    for r in rows:
        k = short(r['Kernel_Name'])
        if not k:
            continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k].add(r['Dispatch_Id'])
    print("==", path)
    for k in agg:
        print("  %-26s dispatches=%5d  " % (k, len(cnt[k])) + "  ".join("%s=%.4g" % kv for kv in sorted(agg[k].items())))
