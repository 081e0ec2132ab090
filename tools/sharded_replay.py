"""Per-rank replay of ONE sharded evaluation on the one GPU there is: rank r of a Pr x Pc grid
ALONE on the GPU behind the replay transport (comm.h: nothing is sent, a receive zero-fills its buffer, an all-reduce is
the identity).  Every launch of the rank's own share of the sweep runs with its real shape, so the wall time of Observe +
Gradient is that rank's COMPUTE time.  Beside it: the bytes the rank would have exchanged (from the layout, DESIGN.md
section 5) at 100 GB/s, and the 1x1 grid of the same code path against the fused single-GPU sweep.

This is a PREDICTION of an N-GPU evaluation's floor -- max over ranks of max(compute, exchange) -- not a measurement of
one: waits for peers' panels, RCCL's own launch costs and link contention are not in it.

usage: python3 tools/sharded_replay.py CONFIG [PrxPc] [N] [out.json]      e.g.  4 2x4 32768 profiles/r05_sharded_replay.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gogp_amd import configs, gp as G, _lib
from gogp_amd.sharded import ShardedGP

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Pr, Pc = (int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "2x4").split("x"))
nobs = int(sys.argv[3]) if len(sys.argv) > 3 else None
out = sys.argv[4] if len(sys.argv) > 4 else None
wl = configs.workload(cfg, nobs, None)
N, D = wl.N, wl.D
prec = 32 if wl.dtype == "f32" else 64
peak = 157.3 if prec == 32 else 78.6
esz = 4 if prec == 32 else 8
X, y = wl.inputs()
world = Pr * Pc
reps = 2 if N > 20000 else 3


def timed(obj):
    obj.Observe(wl.log_theta(0)); obj.Gradient()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(reps):
        obj.Observe(wl.log_theta(1 + k)); obj.Gradient()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


res = {"config": cfg, "N": N, "D": D, "dtype": wl.dtype, "grid": "%dx%d" % (Pr, Pc), "library": _lib.lib().gogp_version().decode(),
       "what": "PREDICTION: per-rank compute time of one sharded Observe + Gradient, each rank alone on one MI355X behind the "
               "replay transport; exchange time = layout bytes / 100 GB/s; neither waits for peers nor RCCL costs are in it"}
g = G.GP(D, wl.simil, wl.noise, X=X, Y=y, precision=prec)
t_fused = timed(g)
g.close()
res["fused_single_gpu_ms"] = t_fused * 1e3
print("fused single-GPU sweep: %.2f ms (%.3f of the roof)" % (t_fused * 1e3, float(N) ** 3 / t_fused / 1e12 / peak), flush=True)
s11 = ShardedGP(D, wl.simil, wl.noise, X=X, Y=y, precision=prec, transport="replay", grid=(1, 1), rank=0, world=1)
t11 = timed(s11)
s11.close()
res["sharded_1x1_ms"] = t11 * 1e3
res["sharded_1x1_over_fused"] = t11 / t_fused
print("sharded code path on a 1x1 grid: %.2f ms = %.3f x the fused sweep (%.3f of the roof)" %
      (t11 * 1e3, t11 / t_fused, float(N) ** 3 / t11 / 1e12 / peak), flush=True)
ranks = []
for r in range(world):
    sh = ShardedGP(D, wl.simil, wl.noise, X=X, Y=y, precision=prec, transport="replay", grid=(Pr, Pc), rank=r, world=world)
    t = timed(sh)
    lb = sh.local_bytes()
    sh.close()
    ranks.append({"rank": r, "pr": r // Pc, "pc": r % Pc, "compute_ms": t * 1e3, "local_bytes": lb})
    print("rank %d (%d, %d): compute %.2f ms" % (r, r // Pc, r % Pc, t * 1e3), flush=True)
npad = -(-N // (512 * Pc)) * 512 * Pc
per_rank_bytes = esz * float(npad) ** 2 * ((Pc - 1) + (Pr - 1)) / (Pr * Pc) + npad / 512 * 512 * 512 * esz
cm = np.array([r["compute_ms"] for r in ranks])
res["ranks"] = ranks
res["compute_ms_max"] = float(cm.max())
res["compute_ms_mean"] = float(cm.mean())
res["ideal_compute_ms"] = t_fused * 1e3 / world
res["per_rank_efficiency_vs_fused"] = float(t_fused * 1e3 / world / cm.max())
res["exchange_bytes_per_rank"] = per_rank_bytes
res["exchange_ms_at_100GBps"] = per_rank_bytes / 100e9 * 1e3
res["predicted_ms_per_evaluation"] = float(max(cm.max(), per_rank_bytes / 100e9 * 1e3))
res["predicted_evals_per_s"] = 1e3 / res["predicted_ms_per_evaluation"]
res["predicted_frac_of_aggregate_roof"] = float(N) ** 3 / (res["predicted_ms_per_evaluation"] * 1e-3) / 1e12 / (peak * world)
print("grid %dx%d: compute max %.2f / mean %.2f ms (ideal fused / %d = %.2f: per-rank efficiency %.2f); exchange %.2f GB per rank = "
      "%.2f ms at 100 GB/s; PREDICTED >= %.2f ms per evaluation = %.2f evals/s = %.2f of the aggregate roof" %
      (Pr, Pc, cm.max(), cm.mean(), world, res["ideal_compute_ms"], res["per_rank_efficiency_vs_fused"], per_rank_bytes / 1e9,
       res["exchange_ms_at_100GBps"], res["predicted_ms_per_evaluation"], res["predicted_evals_per_s"],
       res["predicted_frac_of_aggregate_roof"]), flush=True)
if out:
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
