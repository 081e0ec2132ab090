"""The three recomputable durations behind roofline.frac, from tracked rocprofv3 outputs.

  union      union of the dominant kernel's launch intervals (start / end timestamps of
             `rocprofv3 --kernel-trace`), per evaluation: what bench.py's own HIP events measure
             (roofline.achieved = N^3 / union);
  sum        sum of its launch durations per evaluation: what `--stats` adds up (launches overlap on
             several streams, so sum > wall);
  serialised sum of GRBM_GUI_ACTIVE of its launches / XCDs / clock from the --pmc pass, where kernels
             run one at a time: a lower bound of the achieved rate.

usage: roofline_from_profiles.py <N> <family> <peak TF/s> <kernel-trace dir | .csv> [<pmc dir | .csv>]  -> JSON on stdout
e.g.   python3 tools/roofline_from_profiles.py 16384 dgemm_nt_kernel 78.6 profiles/r03_c3_kernel_trace.csv
"""
import csv, glob, json, sys

N, fam, peak, tdir = float(sys.argv[1]), sys.argv[2], float(sys.argv[3]), sys.argv[4]
pdir = sys.argv[5] if len(sys.argv) > 5 else None
XCDS, CLOCK_HZ = 8, 2.4e9

trace = tdir if tdir.endswith(".csv") else max(glob.glob(tdir + "/*/*_kernel_trace.csv"), key=__import__("os").path.getmtime)
rows = list(csv.DictReader(open(trace)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# an evaluation starts with the split Gram build (two gram_kernel launches)
grams = [e for e in ev if "gram_kernel" in e[2]]
nevals = max(1, len(grams) // 2)
fam_ev = [(s, e) for s, e, n in ev if fam in n]
union, cur_s, cur_e = 0, None, None
for s, e in fam_ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    union += cur_e - cur_s
total = sum(e - s for s, e in fam_ev)
flop = N ** 3
out = {
    "N": int(N), "kernel_family": fam, "evaluations_in_trace": nevals, "launches_per_evaluation": len(fam_ev) / nevals,
    "algorithmic_flop_per_evaluation": flop, "peak_tflops": peak,
    "union_ms_per_evaluation": union / nevals / 1e6, "sum_ms_per_evaluation": total / nevals / 1e6,
    "avg_launch_ms": total / max(1, len(fam_ev)) / 1e6,
}
out["achieved_tflops_union"] = flop / (out["union_ms_per_evaluation"] * 1e-3) / 1e12
out["frac_union"] = out["achieved_tflops_union"] / peak
span = (max(e for _, e, _ in ev) - min(s for s, _, _ in ev)) / 1e6
out["trace_span_ms"] = span
if pdir:
    f = pdir if pdir.endswith(".csv") else max(glob.glob(pdir + "/*/*_counter_collection.csv"), key=__import__("os").path.getmtime)
    gui, disp = 0.0, set()
    for r in csv.DictReader(open(f)):
        if fam in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            gui += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    nev_p = max(1, round(len(disp) / out["launches_per_evaluation"]))
    ser = gui / nev_p / XCDS / CLOCK_HZ * 1e3
    out["serialised_ms_per_evaluation"] = ser
    out["serialised_note"] = "sum of GRBM_GUI_ACTIVE over the family's %d launches (%d evaluations) / %d XCDs / %.1f GHz" % (
        len(disp), nev_p, XCDS, CLOCK_HZ / 1e9)
    out["achieved_tflops_serialised"] = flop / (ser * 1e-3) / 1e12
    out["frac_serialised"] = out["achieved_tflops_serialised"] / peak
print(json.dumps(out, indent=1))
