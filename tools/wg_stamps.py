"""where INSIDE a chain launch the time goes when it runs beside the bulk updates.

Runs ONE Observe (+ Gradient) with the probe build of the library (make -C gogp_amd/csrc stamp -> tools/exp/lib_stamp.so,
-DGOGP_WGSTAMP: every workgroup of the tile kernel and of the diagonal-block kernel writes s_memrealtime at entry /
first operands in LDS / last k-step done / stores drained, plus HW_ID and XCC_ID) and reports, per kernel shape and
stream: how long a launch takes from its first workgroup's entry to its last workgroup's exit, how late its workgroups
START relative to the first (dispatch: waiting for a slot), how long one workgroup RUNS (prologue, k-loop, stores), and
how long the stream waited between the end of its previous launch and the first entry of this one.

usage: python3 tools/wg_stamps.py [N] [D] [eager 0|1] [out.json]      (copies lib_stamp.so over the product library first:
       run it through tools/gpu_stamps.sh, which restores the product library afterwards)"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gogp_amd import gp as G, kernel, synth, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eager = int(sys.argv[3]) if len(sys.argv) > 3 else 1
out = sys.argv[4] if len(sys.argv) > 4 else None
assert "stamp-" in _lib.lib().gogp_version().decode(), "the loaded library is not the probe build"
lib = ctypes.CDLL(_lib.LIB_PATH)
gv = lambda name, ty: ty.in_dll(lib, name)
buf_p = gv("_ZN4gogp11g_stamp_bufE", ctypes.c_void_p)
cap = gv("_ZN4gogp11g_stamp_capE", ctypes.c_longlong)
used = gv("_ZN4gogp12g_stamp_usedE", ctypes.c_longlong)
nrec = gv("_ZN4gogp12g_stamp_nrecE", ctypes.c_longlong)
rec = gv("_ZN4gogp11g_stamp_recE", ctypes.c_longlong * (4 * (1 << 16)))

X, y = synth.make_inputs(N, D, 20251114 + 2)
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
g.set_option("eager", eager)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
for _ in range(2):
    g.Observe(x)
    if eager:
        g.Gradient()
torch.cuda.synchronize()
CAP = 1 << 21
st = torch.zeros(CAP * 8, dtype=torch.int64, device="cuda")
buf_p.value = st.data_ptr()
cap.value = CAP
used.value = 0
nrec.value = 0
g.Observe(x * 1.001)
if eager:
    g.Gradient()
torch.cuda.synchronize()
buf_p.value = None
n = int(nrec.value)
R = np.array(rec[:4 * n], dtype=np.int64).reshape(n, 4)
S = st.cpu().numpy().reshape(CAP, 8)[: int(used.value)].astype(np.float64)
S[:, :4] *= 0.01  # 100 MHz -> us
t00 = S[S[:, 0] > 0, 0].min()
print("N %d eager %d: %d launches, %d workgroups stamped" % (N, eager, n, int(used.value)))
streams = {s: i for i, s in enumerate(sorted(set(R[:, 3])))}
last_end = {}
rows = []
for base, nwg, tag, stream in R:
    s = S[base:base + nwg]
    shape = tag // 10000000000
    if shape == 9:  # diagonal block: entry (0), exit (3)
        ran = s[:, 0] > 0
        if not ran.any():
            continue
        first, end = s[ran, 0].min(), s[ran, 3].max()
        rows.append(dict(kind="diag256", stream=streams[stream], nwg=int(ran.sum()), first=first - t00, span=end - first,
                         gap=first - last_end.get(stream, first), run=float((s[ran, 3] - s[ran, 0]).mean()), start_late=0.0,
                         pro=0.0, kloop=0.0, sto=0.0, K=256, run_min=float((s[ran, 3] - s[ran, 0]).min()), start_p50=0.0))
        last_end[stream] = end
        continue
    ran = s[:, 3] > 0  # workgroups that did not exit early (TRAP / LAUUM tiles outside the triangle)
    if not ran.any():
        continue
    s = s[ran]
    first, end = s[:, 0].min(), s[:, 3].max()
    mode = (tag % 10000000000) // 100000000
    K = ((tag % 100000000) // 100000) * 16
    rows.append(dict(kind={1: "64x64/4w", 2: "128x128/4w", 3: "128x128/8w"}[int(shape)] + " mode%d" % mode, stream=streams[stream],
                     nwg=int(len(s)), first=first - t00, span=end - first, gap=first - last_end.get(stream, first),
                     start_late=float((s[:, 0] - first).max()), start_p50=float(np.median(s[:, 0] - first)),
                     run=float((s[:, 3] - s[:, 0]).mean()), run_min=float((s[:, 3] - s[:, 0]).min()),
                     pro=float((s[:, 1] - s[:, 0]).mean()), kloop=float((s[:, 2] - s[:, 1]).mean()), sto=float((s[:, 3] - s[:, 2]).mean()), K=int(K)))
    last_end[stream] = end
total = max(r["first"] + r["span"] for r in rows)
print("evaluation (first entry -> last exit): %.2f ms" % (total / 1e3))
# which stream is the Cholesky chain: the one that carries the diagonal blocks
chain = next(r["stream"] for r in rows if r["kind"] == "diag256")
print("streams: %s; the Cholesky chain is stream %d" % (sorted(streams.values()), chain))


def summarise(sel, title):
    rr = [r for r in rows if sel(r)]
    if not rr:
        return None
    a = lambda k: np.array([r[k] for r in rr])
    d = dict(title=title, launches=len(rr), nwg_median=float(np.median(a("nwg"))), span_us_median=float(np.median(a("span"))),
             span_us_sum=float(a("span").sum()), gap_before_us_median=float(np.median(a("gap"))), gap_before_us_sum=float(a("gap").sum()),
             last_start_after_first_us_median=float(np.median(a("start_late"))), median_start_after_first_us=float(np.median(a("start_p50"))),
             wg_run_us_mean=float(a("run").mean()), wg_run_us_min=float(a("run_min").min()), wg_prologue_us=float(a("pro").mean()),
             wg_kloop_us=float(a("kloop").mean()), wg_stores_us=float(a("sto").mean()))
    print("%-44s %4d launches  wgs %6.0f  span %7.1f us (sum %8.1f)  gap before %6.1f (sum %8.1f)  last start +%6.1f (median wg +%5.1f)  "
          "wg runs %6.1f (min %5.1f) = pro %5.1f + k %6.1f + st %5.1f" %
          (title, d["launches"], d["nwg_median"], d["span_us_median"], d["span_us_sum"], d["gap_before_us_median"], d["gap_before_us_sum"],
           d["last_start_after_first_us_median"], d["median_start_after_first_us"], d["wg_run_us_mean"], d["wg_run_us_min"],
           d["wg_prologue_us"], d["wg_kloop_us"], d["wg_stores_us"]))
    return d


res = []
res.append(summarise(lambda r: r["kind"] == "diag256", "diag256 (Cholesky chain)"))
for K in sorted(set(r["K"] for r in rows if r["stream"] == chain and r["kind"] != "diag256")):
    for kind in sorted(set(r["kind"] for r in rows if r["stream"] == chain and r["K"] == K and r["kind"] != "diag256")):
        res.append(summarise(lambda r: r["stream"] == chain and r["K"] == K and r["kind"] == kind, "chain  %s K=%d" % (kind, K)))
for st_ in sorted(streams.values()):
    if st_ == chain:
        continue
    for kind in sorted(set(r["kind"] for r in rows if r["stream"] == st_)):
        res.append(summarise(lambda r: r["stream"] == st_ and r["kind"] == kind, "stream %d %s" % (st_, kind)))
# the chain's time line: busy (spans) and waiting (gaps)
cr = [r for r in rows if r["stream"] == chain]
print("Cholesky chain stream: %d launches, sum of spans %.2f ms, sum of gaps %.2f ms, ends at %.2f ms" %
      (len(cr), sum(r["span"] for r in cr) / 1e3, sum(r["gap"] for r in cr) / 1e3, max(r["first"] + r["span"] for r in cr) / 1e3))
if out:
    with open(out, "w") as f:
        json.dump(dict(N=N, D=D, eager=eager, library=_lib.lib().gogp_version().decode(), evaluation_us=total,
                       chain_stream=dict(launches=len(cr), spans_us=sum(r["span"] for r in cr), gaps_us=sum(r["gap"] for r in cr)),
                       groups=[d for d in res if d], launches=rows), f)
