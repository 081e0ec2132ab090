#!/usr/bin/env python3
"""Benchmark of the hot path: GP.Observe + GP.Gradient evaluations per second.

    python bench.py --gpus N --steps K --warmup W [--config C]

N > 1 runs one rank per GPU.  Started WITHOUT a launcher (no WORLD_SIZE in the environment) this
process starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
... bench.py <same arguments>` as a CHILD -- before it has touched the GPU itself -- and relays the one
JSON line and the exit code.  Started by a launcher it checks WORLD_SIZE == N and refuses (exit 2) on a
mismatch: a run never silently degrades to one rank.  Every N > 1 run begins with a pre-flight of the
transport (process group, communicator, one grouped send/recv ring, one all-reduce), each phase under a
30 s watchdog that names rank and phase and exits non-zero.

One "step" = one hyperparameters-only Observe(log theta) (Gram build + blocked
Cholesky + forward solve + LML) followed by Gradient() (triangular inverse,
K^-1, fused gradient reduction), theta changing every step.  X, y are resident
in HBM before the timed region starts.

--config selects the workload from BASELINE.json's `configs` (gogp_amd/configs.py);
default 3 = the configuration the headline metric is quoted on (RBF + white noise,
N=16384, D=8, fp64, one MI355X).

N > 1:
  * configs 1-3 (one evaluation fits and is quoted on ONE GPU): every GPU evaluates its
    own hyperparameter candidate on the full data -- weak scaling, no data-path
    collective; `value` is the replica throughput.  AFTER that measurement the transport
    of the sharded evaluation is pre-flighted (communicator, send/recv ring, all-reduce,
    a first sharded evaluation at N = 64: `preflight`, `rccl_ranks`) and ONE evaluation
    sharded over all ranks is timed and reported as `sharded_evaluation`; a phase of these
    that hangs or fails is reported inside the line and the run exits 0 -- the replica
    measurement does not depend on the library's communicator;
  * configs 4-5 (BASELINE quotes them as ONE evaluation over all GPUs): `value` is the
    throughput of the 2-D block-cyclic sharded evaluation (strong scaling); the pre-flight
    runs first, each phase under a watchdog that ends the run non-zero with rank and phase.

Prints ONE JSON line on rank 0 (contract in the task description), including
  "roofline":     the dominant kernel (fp64 MFMA GEMM/SYRK tile kernel), HIP-event
                  timed on the streams it is launched on during the timed region;
  "cpu_baseline": the CPU oracle's numpy/scipy twin ("port") timed on the host cores, at
                  the FULL size of the workload when N <= 16384 (so that the LML /
                  gradient / mu / sigma errors are measured on the metric's own
                  configuration), on a bounded sample otherwise.
"""
import argparse
import contextlib
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FP64_PEAK_TFLOPS = 78.6  # AMD public spec, fp64 matrix = vector
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32) = vector
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: 8.0 TB/s spec
ORACLE_FULL_N_LIMIT = 16384


def host_cores():
    """CPU threads this process may really use: the cgroup quota if there is one
    (the GPU box shows 256 CPUs but grants 16), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(wl, sample_n, gpu_fn, second_sample_n=0):
    """Time the oracle's fast twin on the first `sample_n` rows of the workload (the whole
    workload when sample_n == N) and compare the GPU path with it on the same rows.

    Every sample reports its phases (Gram pair loop, potrf, potrs, potri, gradient pair loop) and the
    thread pools in effect.  The Cholesky factorisation is timed twice -- LAPACK's threaded dpotrf and
    a dgemm-based blocked one (oracle.potrf_blocked) -- and the faster of the two counts: on the GPU
    boxes (16-CPU quota on a 256-CPU host) OpenBLAS's dpotrf collapses at some sizes (N = 8192 as slow
    as N = 16384), which made round 2's two samples disagree with the N^3 law."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)  # before the C oracle (libgomp) is loaded
    from threadpoolctl import threadpool_info, threadpool_limits
    from oracle.oracle import FastOracle, Oracle  # checker / baseline only
    threadpool_limits(limits=cores)
    N, D = wl.N, wl.D
    X, y = wl.inputs()
    x = wl.log_theta(0)
    Z = wl.test_points(1024 if N >= 1024 else 16)

    def run(n, potrf="lapack", gradient=True, potri="lapack"):
        o = FastOracle(D, wl.simil, wl.noise, block=1024, potrf=potrf, potri=potri)
        o.set_data(X[:n], y[:n])
        t0 = time.time()
        lml = o.Observe(x)
        g = o.Gradient() if gradient else None
        dt = time.time() - t0
        return o, lml, g, dt

    def sample(n):
        """(oracle, lml, grad, seconds with the faster factorisation, report)"""
        o, lml, g, dt = run(n)
        ph = {k: round(v, 4) for k, v in o.timings.items()}
        rep = {"n": n, "seconds_lapack_potrf": dt, "phases_s": ph}
        secs = dt
        if n >= 2048:
            # the dgemm-based twins of the two O(N^3) LAPACK calls; per phase the faster one counts.  (The
            # blocked inverse does 1.5x the flops of dpotri: at N = 16384, where dpotri runs at 660 GFLOP/s
            # on the GPU box's 16 cores, it is only tried when dpotri is slower than 300 GFLOP/s.)
            slow_potri = 2 * float(n) ** 3 / 3 / max(o.timings["potri"], 1e-9) < 300e9
            ob, lml_b, g_b, _ = run(n, potrf="blocked", gradient=slow_potri, potri="blocked")
            pb = ob.timings["potrf"]
            rep["potrf_blocked_s"] = round(pb, 4)
            rep["lml_rel_diff_blocked_vs_lapack"] = abs(lml_b - lml) / abs(lml)
            if pb < o.timings["potrf"]:
                secs -= o.timings["potrf"] - pb
                rep["factorisation_counted"] = "blocked (dgemm-based)"
            else:
                rep["factorisation_counted"] = "LAPACK dpotrf"
            rep["inverse_counted"] = "LAPACK dpotri"
            if slow_potri:
                qb = ob.timings["potri"]
                rep["potri_blocked_s"] = round(qb, 4)
                rep["grad_rel_diff_blocked_vs_lapack"] = float(np.abs(g_b - g).max() / max(1.0, np.abs(g).max()))
                if qb < o.timings["potri"]:
                    secs -= o.timings["potri"] - qb
                    rep["inverse_counted"] = "blocked (dgemm-based)"
            del ob
        rep["seconds"] = secs
        flop = float(n) ** 3
        rep["gflops"] = {"potrf_lapack": round(flop / 3 / max(o.timings["potrf"], 1e-9) / 1e9, 1),
                         "potri": round(2 * flop / 3 / max(o.timings["potri"], 1e-9) / 1e9, 1)}
        if "potrf_blocked_s" in rep:
            rep["gflops"]["potrf_blocked"] = round(flop / 3 / max(rep["potrf_blocked_s"], 1e-9) / 1e9, 1)
        if "potri_blocked_s" in rep:  # useful flop of the inverse (the blocked twin executes 1.5x as many)
            rep["gflops"]["potri_blocked"] = round(2 * flop / 3 / max(rep["potri_blocked_s"], 1e-9) / 1e9, 1)
        return o, lml, g, secs, rep

    # untimed warm-up: thread pools, first-touch page-in of OpenBLAS / libgomp
    run(min(1024, sample_n))
    o, lml, g, dt, rep = sample(sample_n)
    full = sample_n == N
    scale = (sample_n / float(N)) ** 3
    out = {
        "value": (1.0 / dt) * scale,
        "unit": "evals/s",
        "cores": int(cores),
        "kind": "port",
        "sample": ("1 Observe+Gradient at N=%d D=%d (%s), %.1f s of scipy/OpenBLAS potrf+potri+potrs and "
                   "C/OpenMP Gram + gradient pair loops%s"
                   % (sample_n, D, "the FULL workload" if full else "first rows of the same inputs", dt,
                      "" if full else "; scaled by (%d/%d)^3 to N=%d" % (sample_n, N, N))),
        "measured_at_full_n": bool(full),
        "seconds": dt,
        "detail": rep,
        "thread_pools": [{"api": i.get("internal_api"), "threads": i.get("num_threads"),
                          "layer": i.get("threading_layer"), "lib": os.path.basename(i.get("filepath", ""))}
                         for i in threadpool_info()],
    }
    mu, sigma = o.Produce(Z)
    lml_gpu, g_gpu, mu_gpu, sigma_gpu = gpu_fn(X[:sample_n], y[:sample_n], x, Z)
    errs = {
        "n": sample_n,
        "lml_rel_err_vs_oracle": abs(lml_gpu - lml) / abs(lml),
        "grad_rel_err_vs_oracle": float(np.abs(g_gpu - g).max() / max(1.0, np.abs(g).max())),
        "mu_rel_err_vs_oracle": float(np.abs(mu_gpu - mu).max() / max(1e-300, np.abs(mu).max())),
        "sigma_rel_err_vs_oracle": float(np.abs(sigma_gpu - sigma).max() / max(1e-300, np.abs(sigma).max())),
    }
    del o
    if second_sample_n and second_sample_n < sample_n:
        _, _, _, dt2, rep2 = sample(second_sample_n)
        ratio = (dt / dt2) / (sample_n / float(second_sample_n)) ** 3
        out["second_sample"] = {"n": second_sample_n, "seconds": dt2, "detail": rep2,
                                "evals_per_s_scaled_to_N": (1.0 / dt2) * (second_sample_n / float(N)) ** 3,
                                "time_ratio_over_n3_law": ratio,
                                "note": "time(N) / time(n) divided by (N/n)^3: 1 = the N^3 law.  OpenBLAS's threaded "
                                        "dpotrf / dpotri collapse at N = 8192 on the GPU boxes (16-CPU quota on a "
                                        "256-CPU host: 32 and 75 GFLOP/s against 247 and 663 at N = 16384), so per "
                                        "phase the faster of LAPACK and a dgemm-based blocked twin counts; what is "
                                        "left of the gap is the rate of dgemm itself at the two sizes (gflops)"}
    # the reference's own algorithm (dense dK per parameter, r0 = aa^T dK, r1 = K^-1 dK:
    # gp/gp.go:476-485; 4P N^3 flop) as restated by the faithful C oracle, single thread,
    # at a size it finishes in about a second, extrapolated by its N^3 law
    nf = min(512 if wl.P <= 4 else 256, sample_n)
    of = Oracle(D, wl.simil, wl.noise)
    of.set_data(X[:nf], y[:nf])
    t0 = time.time()
    of.Observe(x)
    of.Gradient()
    dtf = time.time() - t0
    out["faithful_algorithm"] = {
        "n": nf, "cores": 1, "seconds": dtf,
        "extrapolated_evals_per_s": (1.0 / dtf) * (nf / float(N)) ** 3,
        "note": "dense-dK algorithm of gp/gp.go:418-499 in C (no AD tape, no Go runtime), "
                "scaled by (n/N)^3"}
    return out, errs


def pmc_traffic(config, build_id):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary of the same command
    (profiles/r04_pmc_traffic.json, written by tools/pmc_traffic.py from separate rocprofv3 --pmc passes,
    FETCH_SIZE doubled per the gfx950 correction).  The file is stamped with the build id of the library
    it was measured on (gogp_version(): hash of the library's sources); with another build the number is
    stale and is NOT reported.  Returns (entry or None, reason)."""
    d = None
    for name in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):  # the newest round's file that exists
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            break
        except Exception:
            continue
    if d is None:
        return None, "no profiles/r0*_pmc_traffic.json"
    e = d.get(str(config))
    if e is None:
        return None, "no PMC pass for this configuration"
    if d.get("build") != build_id:
        return None, "stale: measured on build %s, this library is build %s" % (d.get("build"), build_id)
    return e, None


def launch_ranks(ngpus, argv):
    """`--gpus N` without a launcher: one rank per GPU through torch.distributed.run, started as a child
    process BEFORE this process makes any GPU call (a process that has initialised the GPU must not
    start or replace another on this pool).  The ranks inherit stdout / stderr: rank 0's JSON line is
    this run's line; the child's exit code is this run's exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it here
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def collective_timeout(line, sharded_value, rank, seconds, json_out, measurement_done=None):
    """What a rank does when the watchdog over the collective part of an N > 1 run expires; returns the
    exit code.  Replica configuration with the measurement complete: `value` (independent evaluations, no
    data-path collective) stands; what hangs is the ADDITIONAL evaluation sharded over all ranks -- rank 0
    prints the line with the extra marked as timed out, every rank exits 0.  Otherwise (the sharded
    evaluation IS the value, or nothing was measured yet): rank 0 prints what it has with `error`, exit 3.
    `measurement_done` is state EVERY rank has (set after the replica timing); only rank 0 holds a line, so
    deciding from `line` alone would make ranks 1.. exit 3 and the launcher tear rank 0 down before it prints."""
    done = (line is not None) if measurement_done is None else bool(measurement_done)
    if done and not sharded_value:
        if rank == 0 and line is not None:
            line["transport_ok"] = False
            line["sharded_evaluation"] = {
                "error": "timed out after %d s (the line's value is the replica measurement, taken before "
                         "this additional sharded evaluation)" % seconds}
            print(json.dumps(line), file=json_out, flush=True)
        return 0
    if rank == 0:
        line = line or {"metric": "GP.Observe+Gradient evals/sec", "value": None}
        line["error"] = "collective timed out after %d s" % seconds
        print(json.dumps(line), file=json_out, flush=True)
    return 3


class PhaseWatchdog:
    """`with wd.phase("name", seconds):` -- if the block does not finish in time, say which rank is stuck
    in which phase (stderr; rank 0 also prints a JSON line carrying the error) and end the process
    non-zero.  The collective calls block in C with the GIL released, so the timer thread gets to run."""

    def __init__(self, rank, json_out, line_holder):
        self.rank, self.json_out, self.line_holder = rank, json_out, line_holder
        #: replica configuration with the measurement complete (the pre-flight of the sharded transport
        #: then runs AFTER it): an expiry is reported inside the line (`preflight.error`) and the ranks
        #: exit 0 -- the replica measurement needs no communicator of the library
        self.soft = False

    def _bail(self, name, seconds):
        msg = "rank %d: phase '%s' did not finish within %d s" % (self.rank, name, seconds)
        print("bench.py: " + msg, file=sys.stderr, flush=True)
        if self.soft:
            if self.rank == 0 and self.line_holder.get("line") is not None:
                line = self.line_holder["line"]
                line["preflight"] = {"error": msg}
                line["transport_ok"] = False
                line["rccl_ranks"] = 0
                line["sharded_evaluation"] = {"error": "not run: the transport's pre-flight did not finish"}
                try:
                    print(json.dumps(line), file=self.json_out, flush=True)
                except Exception:
                    pass
            os._exit(0)
        if self.rank == 0:
            line = self.line_holder.get("line") or {"metric": "GP.Observe+Gradient evals/sec", "value": None}
            line["error"] = msg
            line["transport_ok"] = False
            line["rccl_ranks"] = 0
            try:
                print(json.dumps(line), file=self.json_out, flush=True)
            except Exception:
                pass
        os._exit(4)

    @contextlib.contextmanager
    def phase(self, name, seconds=30):
        t = threading.Timer(seconds, self._bail, (name, seconds))
        t.daemon = True
        t.start()
        try:
            yield
        finally:
            t.cancel()


def stub_rank_body(args, rank, world, json_out):
    """Test double of the rank body (tests/test_bench_launcher.py, GOGP_BENCH_STUB=1): the ranks meet
    over gloo on the CPU, rank 0 prints a line with what it saw, every rank exits with the requested
    code.  Nothing here touches the GPU or the library."""
    import torch
    import torch.distributed as dist
    seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
    if rank == 0:
        print(json.dumps({"stub": True, "n_gpus": world, "ranks_seen": seen, "gpus_arg": args.gpus,
                          "config": args.config, "steps": args.steps, "nobs": args.nobs}),
              file=json_out, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    code = int(os.environ.get("GOGP_BENCH_STUB_EXIT", "0"))
    if code and rank == int(os.environ.get("GOGP_BENCH_STUB_EXIT_RANK", "0")):
        sys.exit(code)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=3, choices=[1, 2, 3, 4, 5])
    ap.add_argument("--nobs", type=int, default=None)
    ap.add_argument("--ndim", type=int, default=None)
    ap.add_argument("--cpu-sample-n", type=int, default=8192,
                    help="CPU baseline sample when the full workload is too large for the oracle")
    ap.add_argument("--no-produce", action="store_true",
                    help="skip the secondary Produce measurement (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--candidates", type=int, default=4,
                    help="N <= 8192: also time this many candidates evaluated concurrently")
    ap.add_argument("--candidates-per-step", type=int, default=None,
                    help="candidate thetas evaluated per step in ONE launch sequence "
                         "(gogp_observe_gradient_candidates); default 8 for configs 1 and 2 (N <= 4096: one "
                         "evaluation is a latency-bound chain), 1 otherwise")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="gogp_set_option on the benchmarked handle (A/B runs), e.g. superpanel=3")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--sharded", action="store_true",
                    help="--gpus 1: also time the SHARDED code path (dist2d.hip) on a 1x1 grid over the RCCL transport "
                         "and report it beside the fused sweep as `sharded_1x1` (default for configs 4 and 5)")
    ap.add_argument("--sharded-timeout", type=int, default=300)
    ap.add_argument("--preflight-timeout", type=int, default=30,
                    help="seconds per pre-flight phase of an N > 1 run before the watchdog ends it")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: start the N ranks ourselves, as a child, before any GPU call
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: launched with WORLD_SIZE=%d but --gpus %d: refusing to run (the line would not "
              "describe the run that was asked for)" % (world, args.gpus), file=sys.stderr, flush=True)
        sys.exit(2)
    # stdout carries exactly ONE line (the JSON): whatever libraries print there (gloo's
    # connection notes, RCCL's version banner) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(json_fd, "w")
    if os.environ.get("GOGP_BENCH_STUB"):
        stub_rank_body(args, rank, world, json_out)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # one process per GPU; GOGP_DIST_BACKEND=gloo lets several ranks share one GPU for
    # rehearsals on a 1-GPU box (RCCL refuses duplicate devices)
    backend = os.environ.get("GOGP_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    from gogp_amd import configs
    from gogp_amd import dist as gd
    from gogp_amd import gp as G
    out_holder = {"line": None}
    pw = PhaseWatchdog(rank, json_out, out_holder)
    with pw.phase("process group init (%s)" % backend, max(args.preflight_timeout, 60)):
        gd.init(backend, torch.device("cuda", local_rank))
    red_dev = "cuda" if backend == "nccl" else "cpu"

    wl = configs.workload(args.config, args.nobs, args.ndim)
    # ---- N > 1: pre-flight of the transport the sharded evaluation uses --------------------------------
    # Where the sharded evaluation IS the value (configs 4, 5) it runs before anything is timed and a phase
    # that does not finish ends the run non-zero.  In a replica configuration (1-3) the value needs no
    # communicator of the library: it is measured first, the pre-flight and the additional sharded evaluation
    # follow, and a failure there is reported inside the line.
    def run_preflight():
        from gogp_amd.sharded import ShardedGP
        Xp, yp = wl.inputs()
        Xp, yp = Xp[:64], yp[:64]
        tp0 = time.perf_counter()
        with pw.phase("communicator init (unique id broadcast + ncclCommInitRank)", args.preflight_timeout):
            sgp = ShardedGP(wl.D, wl.simil, wl.noise, X=Xp, Y=yp, device=local_rank)
        tp1 = time.perf_counter()
        with pw.phase("grouped send/recv ring (rank -> rank+1)", args.preflight_timeout):
            sgp.selftest(0)
        tp2 = time.perf_counter()
        with pw.phase("all-reduce", args.preflight_timeout):
            sgp.selftest(1)
        tp3 = time.perf_counter()
        with pw.phase("first sharded evaluation (N = 64)", args.preflight_timeout):
            sgp.Observe(wl.log_theta(0))
            sgp.Gradient()
        nranks_comm, is_rccl = sgp.comm_ranks()
        pf = {"comm_ranks": nranks_comm, "rccl": is_rccl, "transport": sgp.transport_text(),
              "grid": sgp.grid_text(), "comm_init_s": tp1 - tp0, "ring_s": tp2 - tp1,
              "allreduce_s": tp3 - tp2, "first_eval_s": time.perf_counter() - tp3}
        sgp.close()
        if nranks_comm != world:
            raise RuntimeError("the communicator counts %d ranks, the launcher %d" % (nranks_comm, world))
        return pf

    preflight = None
    if world > 1 and wl.sharded:
        try:
            preflight = run_preflight()
        except RuntimeError as e:
            print("bench.py: rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
            os._exit(5)
    N, D = wl.N, wl.D
    big = N > 20000
    steps = args.steps if args.steps is not None else (3 if N > 40000 else 5 if big else 10 if N > 6000 else 50)
    warmup = args.warmup if args.warmup is not None else (1 if big else 2)
    sharded_value = wl.sharded and world > 1  # `value` = the sharded evaluation
    # configs[4] is an fp32 configuration: option precision = 32 on one GPU and on the shards alike
    # (float tiles and exchanges, fp32 MFMA; fp64 diagonal tiles, reductions and refinement of alpha)
    prec = 32 if wl.dtype == "f32" else 64
    dtype = "f32" if prec == 32 else "f64"
    peak = FP32_PEAK_TFLOPS if prec == 32 else FP64_PEAK_TFLOPS
    esz = 4.0 if prec == 32 else 8.0  # bytes per element of the N x N matrices
    X, y = wl.inputs()
    simil, noise = wl.simil, wl.noise
    cps = args.candidates_per_step
    if cps is None:
        cps = 8 if (args.config in (1, 2) and prec == 64 and not sharded_value) else 1
    if cps > 1 and (prec != 64 or sharded_value):
        raise SystemExit("--candidates-per-step > 1 needs the fp64 single-GPU path")

    def sync():
        gd.barrier()
        torch.cuda.synchronize()

    # Watchdog for everything that involves a collective on a path that cannot be rehearsed on
    # real multi-GPU RCCL before the driver runs it: if it hangs, rank 0 prints what it has
    # and EVERY rank exits non-zero, so the driver records the hang as a failure.
    wd = None
    if world > 1:
        def _bail():
            os._exit(collective_timeout(out_holder["line"], sharded_value, rank, args.sharded_timeout, json_out,
                                        measurement_done=out_holder.get("measurement_done", False)))

        wd = threading.Timer(args.sharded_timeout, _bail)
        wd.daemon = True
        wd.start()

    out = None
    g = None
    lbfgs_sharded = None
    if not sharded_value:
        g = G.GP(D, simil, noise, device=local_rank, precision=prec)
        for ov in args.option:
            g.set_option(ov.split("=")[0], int(ov.split("=")[1]))
        # inputs resident in HBM before anything is timed
        dX = torch.from_numpy(X).to("cuda")
        dy = torch.from_numpy(y).to("cuda")
        torch.cuda.synchronize()
        g.set_data_device(dX.data_ptr(), dy.data_ptr(), N)

        def step(k):
            if cps > 1:  # cps candidate thetas in one launch sequence
                xs = np.array([wl.log_theta(k * cps + i, rank) for i in range(cps)])
                lmls, grads, _ = g.observe_gradient_candidates(xs)
                return float(lmls[-1]), grads[-1]
            lml = g.Observe(wl.log_theta(k, rank))
            grad = g.Gradient()
            return lml, grad

        for k in range(warmup):
            step(k)
        g.profile_enable(True)
        sync()
        t0 = time.perf_counter()
        for k in range(steps):
            lml, grad = step(warmup + k)
        sync()
        dt = time.perf_counter() - t0
        gemm_ms, gemm_launches, gemm_flops, gemm_busy_ms = g.profile_read()
        gram_ms, gram_n = g.profile_read_aux(0)
        grad_ms, grad_n = g.profile_read_aux(1)
        g.profile_enable(False)
        dt = gd.max_over_ranks(dt, device=red_dev)
        value = world * steps * cps / dt
        par_text = ("1 evaluation per GPU" if world == 1 else
                    "replicas: %d independent evaluations (one candidate theta per GPU)" % world)
        if cps > 1:
            par_text = ("%d candidate thetas per step in ONE launch sequence per GPU (candidate index on the "
                        "grid's z axis; gogp_observe_gradient_candidates)" % cps) + (
                        "" if world == 1 else "; replicas: %d GPUs, independent batches" % world)
        scaling = "weak"
    else:
        from gogp_amd.sharded import ShardedGP
        sg = ShardedGP(D, simil, noise, X=X, Y=y, device=local_rank, precision=prec)

        def step(k):
            lml = sg.Observe(wl.log_theta(k))  # the same theta on every rank: ONE evaluation
            grad = sg.Gradient()
            return lml, grad

        for k in range(warmup):
            step(k)
        sg.profile_enable(True)
        sync()
        t0 = time.perf_counter()
        for k in range(steps):
            lml, grad = step(warmup + k)
        sync()
        dt = time.perf_counter() - t0
        gemm_ms, gemm_launches, gemm_flops, gemm_busy_ms = sg.profile_read()
        gram_ms = gram_n = grad_ms = grad_n = 0
        sg.profile_enable(False)
        dt = gd.max_over_ranks(dt, device=red_dev)
        value = steps / dt
        par_text = "ONE evaluation sharded 2-D block-cyclically over a %s grid of GPUs (%s)" % (
            sg.grid_text(), sg.transport_text())
        scaling = "strong"
        if not args.no_produce:
            # configs[4]: "LML+grad inside L-BFGS hyperparameter loop" -- the optimiser over the SHARDED
            # handle: every rank runs the same (deterministic) iteration on the same all-reduced LML and
            # gradient, so the ranks stay in step without any extra exchange
            from gogp_amd import optimize
            iters = 2 if N > 40000 else 4
            sync()
            tl = time.perf_counter()
            try:
                res = optimize.lbfgs(sg, wl.log_theta(0), major_iterations=iters, gradient_threshold=1e-9)
                sync()
                tl = time.perf_counter() - tl
                lbfgs_sharded = {
                    "major_iterations": res.iterations, "evaluations": res.evaluations, "seconds": tl,
                    "evals_per_s": res.evaluations / tl, "lml_start": res.history[0], "lml_end": res.lml,
                    "note": "optimize.lbfgs over the sharded handle, the same iterate sequence on every rank"}
                # k = 2 candidates per call on the shards (gogp_observe_gradient_candidates: evaluated one after the
                # other in the shards' own tiles): against two single calls, bit for bit
                xs2 = np.stack([wl.log_theta(1), wl.log_theta(2)])
                single = [(sg.Observe(x), sg.Gradient()) for x in xs2]
                sync()
                tc = time.perf_counter()
                lm2, gr2, st2 = sg.observe_gradient_candidates(xs2)
                sync()
                tc = time.perf_counter() - tc
                lbfgs_sharded["candidates_k2"] = {
                    "seconds": tc, "evals_per_s": 2.0 / tc, "status": [int(v) for v in st2],
                    "bit_identical_to_single_calls": bool(all(lm2[c] == single[c][0] and np.array_equal(gr2[c], single[c][1])
                                                              for c in range(2)))}
            except Exception as e:  # noqa: BLE001
                lbfgs_sharded = {"error": repr(e)[:200]}

    if rank == 0:
        algo_flops_step = float(N) ** 3 * cps  # N^3/3 Cholesky + 2N^3/3 inverse per evaluation
        # The kernel's launches overlap (several streams): its busy time is the union of the
        # event-timed launch intervals, not their sum.  On a sharded run the counters are rank
        # 0's and the algorithmic work per rank is N^3 / world.
        per_rank = algo_flops_step / (world if sharded_value else 1)
        achieved = (per_rank * steps / (gemm_busy_ms * 1e-3) / 1e12 if gemm_busy_ms > 0 else 0.0)
        try:
            peak_cal = G.mfma_f64_peak(20000, local_rank) if prec == 64 else None
            peak_cal32 = G.mfma_f32_peak(20000, local_rank) if prec == 32 else None
        except Exception:
            peak_cal = peak_cal32 = None
        out = {
            "metric": "GP.Observe+Gradient evals/sec (%s) at N=%d D=%d" % (
                "fp64" if dtype == "f64" else "fp32", N, D),
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {
                "workload": wl.name + ", Observe+Gradient (hyperparameters-only form), theta perturbed "
                                      "every step",
                "baseline_config": wl.config, "N": N, "D": D, "kernel": wl.kernel_text, "P": wl.P,
                "candidates_per_step": cps,
                "parallelism": par_text,
            },
            "lml": lml,
            # N > 1: did the transport of the sharded evaluation come up (communicator, send/recv ring, all-reduce,
            # a first sharded evaluation)?  One boolean a driver can read without parsing `preflight`; rccl_ranks is
            # ncclCommCount, 0 (never null) when RCCL did not carry the run.  N = 1: no transport, both absent.
            **({"transport_ok": bool(preflight and "error" not in preflight),
                "rccl_ranks": int(preflight["comm_ranks"]) if preflight and preflight.get("rccl") else 0}
               if world > 1 else {}),
            "preflight": preflight,
            "roofline": {
                "bound": "mfma",
                "kernel": ("gogp::sgemm_nt_kernel (v_mfma_f32_32x32x2_f32 GEMM/SYRK tile kernel)" if prec == 32 else
                           "gogp::dgemm_nt_kernel (v_mfma_f64_16x16x4_f64 GEMM/SYRK tile kernel)"),
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "frac_wall": (per_rank / (dt / steps) / 1e12) / peak,
                "traffic": None,
                "algorithmic_flops_per_step": algo_flops_step,
                "launches_per_step": gemm_launches / max(1, steps),
                "avg_launch_ms": gemm_ms / max(1, gemm_launches),
                "kernel_busy_ms_per_step": gemm_busy_ms / max(1, steps),
                "sum_of_launch_durations_ms_per_step": gemm_ms / max(1, steps),
                "launch_concurrency": gemm_ms / gemm_busy_ms if gemm_busy_ms > 0 else None,
                "launched_flops_per_step": gemm_flops / max(1, steps),
                "peak_calibrated_mfma_f64": peak_cal,
                "peak_calibrated_mfma_f32": peak_cal32,
                "note": "achieved = N^3 algorithmic flop per step / HIP-event-timed busy time of the "
                        "kernel per step (union of its launch intervals: launches overlap on several "
                        "streams; rocprofv3 --stats sums them, see sum_of_launch_durations_ms_per_step = "
                        "avg_launch_ms x launches_per_step); frac_wall = the same flop / wall time per "
                        "step (a lower bound that needs no event arithmetic); peak = 78.6 TFLOP/s fp64 spec "
                        "(157.3 fp32 matrix); peak_calibrated = sustained v_mfma_f64 (fp32 path: v_mfma_f32_32x32x2) "
                        "issue-rate microbenchmark on this device",
            },
        }
        version = G._lib.lib().gogp_version().decode()
        build_id = version.split("build ")[-1] if "build " in version else "unknown"
        out["library"] = version
        tr, why = pmc_traffic(wl.config, build_id)
        if tr is not None and tr.get("N") == N and not sharded_value and tr.get("candidates_per_step", 1) == cps:
            out["roofline"]["traffic"] = tr.get("bytes_per_launch")
            out["roofline"]["traffic_note"] = tr.get("note")
        else:
            out["roofline"]["traffic_note"] = "not reported: " + (why or "measured at another size / on one GPU only")
        if gram_n and grad_n:
            w = 512  # the first super-panel's block columns are built on the panel stream
            # bytes per matrix element: 8 (fp64 path) or 4 (precision = 32: K, K^-1 are float) -- round 3 counted 8
            # on both paths and overstated config 5's GB/s twofold
            gram_bytes = esz * max(0, N - w) ** 2 / 2.0 * cps
            grad_bytes = esz * float(N) * N / 2.0 * cps
            out["hbm_bound_kernels"] = {
                "gram_build": {"algorithmic_bytes": gram_bytes, "ms": gram_ms / gram_n,
                               "GBps": gram_bytes / (gram_ms / gram_n * 1e-3) / 1e9,
                               "frac_of_hbm_peak": gram_bytes / (gram_ms / gram_n * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "grad_reduce": {"algorithmic_bytes": grad_bytes, "ms": grad_ms / grad_n,
                                "GBps": grad_bytes / (grad_ms / grad_n * 1e-3) / 1e9,
                                "frac_of_hbm_peak": grad_bytes / (grad_ms / grad_n * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "bytes_per_element": esz,
                "note": "8 B (fp64 path; 4 B with precision = 32) per lower-triangle element written (Gram; the part built on the main "
                        "stream) / read (K^-1 in the fused gradient reduction); HIP events on the "
                        "kernel's stream"}
        if cps > 1 and N <= 1024:
            # up to N = 1024 the candidates' launch sequence is replayed from a captured hipGraph --
            # except while the tile kernel's launches carry profiling events, as in the timed region
            # above: the same steps again without instrumentation
            step(0); step(1)
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for kk in range(steps):
                step(2 + kk)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            out["hipgraph_replay"] = {
                "evals_per_s": steps * cps / tg, "ms_per_step": tg / steps * 1e3,
                "note": "the timed steps repeated with profiling events off: the launch sequence of a step "
                        "is one captured hipGraph (linear; N <= 1024)"}
        if cps > 1 and 1024 < N <= 8192:
            # the same steps replayed from an EXPLICITLY built hipGraph (option graph = 2: one node per launch / copy,
            # the sweep's cross-stream dependencies as edges, no stream capture -- graphrec.h).  Reported, not used for
            # `value`: this runtime executes the graph's parallel branches no faster than their serialisation.
            ref_l, ref_g = step(3)
            g.set_option("graph", 2)
            step(0); step(1); step(2)
            got_l, got_g = step(3)
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for kk in range(steps):
                step(2 + kk)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            nodes, refused = g.graph_info()
            g.set_option("graph", 1)
            out["hipgraph_replay"] = {
                "evals_per_s": steps * cps / tg, "ms_per_step": tg / steps * 1e3, "nodes": nodes, "refused_by_runtime": refused,
                "bit_identical_to_stream_path": bool(ref_l == got_l and np.array_equal(ref_g, got_g)),
                "note": "explicitly built hipGraph of one step (hipGraphAddKernelNode / MemcpyNode / MemsetNode with "
                        "explicit dependencies); `value` above is the stream path, which is faster on this runtime"}
        if cps > 1:
            # the same workload one candidate at a time (the latency-bound chain `value` amortises)
            g.Observe(wl.log_theta(0)); g.Gradient()
            torch.cuda.synchronize()
            reps = max(5, steps)
            t1 = time.perf_counter()
            for r in range(reps):
                g.Observe(wl.log_theta(r)); g.Gradient()
            torch.cuda.synchronize()
            t1 = (time.perf_counter() - t1) / reps
            out["single_candidate"] = {
                "evals_per_s": 1.0 / t1, "ms_per_eval": t1 * 1e3,
                "frac_wall": float(N) ** 3 / t1 / 1e12 / peak,
                "note": "Observe + Gradient one theta at a time on the same handle (what a strictly "
                        "sequential optimiser sees); `value` evaluates candidates_per_step thetas per "
                        "launch sequence"}
        if world == 1 and N <= 8192 and args.candidates > 1:
            # below N ~ 8192 one evaluation is a chain of small dependent launches: k candidates
            # evaluated at once (gogp_observe_gradient_batch; the reference's optimiser can do the
            # same, optimize.Settings.Concurrent, tutorial/tutorial.go:141) overlap their chains
            k = args.candidates
            gps = [g] + [G.GP(D, simil, noise, device=local_rank, precision=prec) for _ in range(k - 1)]
            for gg in gps[1:]:
                gg.set_data_device(dX.data_ptr(), dy.data_ptr(), N)
            xs = np.array([wl.log_theta(i) for i in range(k)])
            G.observe_gradient_batch(gps, xs)
            torch.cuda.synchronize()
            reps = max(5, steps // k)
            tb = time.perf_counter()
            for r in range(reps):
                xs = np.array([wl.log_theta(r * k + i) for i in range(k)])
                G.observe_gradient_batch(gps, xs)
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
            out["concurrent_candidates"] = {
                "k": k, "evals_per_s": reps * k / tb, "ms_per_batch": tb / reps * 1e3,
                "note": "k independent candidates evaluated at once on this GPU, one handle and one "
                        "host thread each (gogp_observe_gradient_batch), for comparison with the "
                        "one-launch-sequence form"}
            for gg in gps[1:]:
                gg.close()
        if world == 1 and prec == 64 and cps == 1 and N >= 4096 and g is not None and not args.no_produce:
            # beyond the native-fp64 roof with LML / mu / sigma untouched -- "compute only the
            # gradient's K^-1 in fp32 MFMA".  Option gradient_precision = 32: fp64 factorisation, LML, alpha, Produce;
            # Y = L^-T and K^-1 = Y Y^T on the fp32 tile kernel.  Reported beside `value`, never as it.
            lml_n = g.Observe(wl.log_theta(0)); grad_n = g.Gradient()
            g.set_option("gradient_precision", 32)
            lml_m = g.Observe(wl.log_theta(0)); grad_m = g.Gradient()
            torch.cuda.synchronize()
            reps = max(3, steps // 2)
            tm = time.perf_counter()
            for r in range(reps):
                g.Observe(wl.log_theta(1 + r)); g.Gradient()
            torch.cuda.synchronize()
            tm = (time.perf_counter() - tm) / reps
            g.set_option("gradient_precision", 64)
            out["mixed_precision_gradient"] = {
                "evals_per_s": 1.0 / tm, "ms_per_eval": tm * 1e3,
                "lml_identical_to_native": bool(lml_m == lml_n),
                "grad_rel_diff_vs_native": float(np.abs(grad_m - grad_n).max() / max(1e-300, np.abs(grad_n).max())),
                "note": "option gradient_precision = 32 (not the default, not `value`): fp64 Cholesky / LML / alpha / "
                        "Produce, the triangular inverse and K^-1 = Y Y^T in fp32 from a float copy of the fp64 factor, "
                        "trace and output-scale components of the gradient from closed forms; the reference checks "
                        "its gradient to 1e-4 (gp_test.go:170,248)"}
        if world == 1 and not args.no_produce and g is not None:
            # configs[4] words the workload as "LML+grad inside L-BFGS hyperparameter loop": a few major
            # iterations of the optimiser on this handle (gogp_amd/optimize.py) -- the same
            # evaluations, now chosen by the line search
            from gogp_amd import optimize
            iters = 2 if N > 40000 else 4
            torch.cuda.synchronize()
            tl = time.perf_counter()
            try:
                res = optimize.lbfgs(g, wl.log_theta(0), major_iterations=iters, gradient_threshold=1e-9)
                torch.cuda.synchronize()
                tl = time.perf_counter() - tl
                out["lbfgs_loop"] = {
                    "major_iterations": res.iterations, "evaluations": res.evaluations, "seconds": tl,
                    "evals_per_s": res.evaluations / tl, "lml_start": res.history[0], "lml_end": res.lml,
                    "note": "optimize.lbfgs from the first theta: every trial point is one Observe (+ Gradient "
                            "when accepted) on the resident data"}
            except Exception as e:  # noqa: BLE001
                out["lbfgs_loop"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_produce and g is not None and N <= 40000:
            # the boundary's host-buffer form (gogp_set_data: X, y from host memory): the same
            # evaluation with the inputs re-sent over PCIe every step -- never `value`
            reps = 3 if N > 6000 else 10
            g.X, g.Y = X, y
            g.Observe(wl.log_theta(0)); g.Gradient()
            tpc = time.perf_counter()
            for r in range(reps):
                g.X, g.Y = X, y  # marks the data dirty: the next Observe uploads them again
                g.Observe(wl.log_theta(r)); g.Gradient()
            torch.cuda.synchronize()
            tpc = (time.perf_counter() - tpc) / reps
            out["pcie_inclusive"] = {
                "evals_per_s": 1.0 / tpc, "ms_per_eval": tpc * 1e3, "host_bytes_per_eval": 8.0 * N * (D + 1),
                "note": "X (N x D) and y re-uploaded from pageable host memory before every evaluation "
                        "(gogp_set_data drains the streams and copies synchronously), one candidate at a time"}
            g.set_data_device(dX.data_ptr(), dy.data_ptr(), N)
            g.Observe(wl.log_theta(0))  # new data invalidate the factorisation: Produce below needs one
        if world == 1 and g is not None and (args.sharded or (wl.sharded and not args.no_sharded)) and cps == 1:
            # the code path configs[3] / configs[4] run on 8 GPUs, on the one GPU there is: a 1x1 grid of the 2-D
            # block-cyclic sweep over the RCCL transport (one rank: communicator, all-reduces, every launch and filter of
            # dist2d.hip) beside the fused single-GPU sweep -- its per-rank efficiency
            try:
                from gogp_amd.sharded import ShardedGP
                s11 = ShardedGP(D, simil, noise, X=X, Y=y, device=local_rank, precision=prec, grid=(1, 1))
                s11.Observe(wl.log_theta(0)); s11.Gradient()
                torch.cuda.synchronize()
                r11 = 2 if N > 40000 else 3
                t11 = time.perf_counter()
                for r in range(r11):
                    l11 = s11.Observe(wl.log_theta(1 + r)); g11 = s11.Gradient()
                torch.cuda.synchronize()
                t11 = (time.perf_counter() - t11) / r11
                lf = g.Observe(wl.log_theta(r11)); gf = g.Gradient()
                out["sharded_1x1"] = {
                    "ms_per_eval": t11 * 1e3, "evals_per_s": 1.0 / t11, "frac_wall": float(N) ** 3 / t11 / 1e12 / peak,
                    "over_fused_sweep": t11 / (dt / steps), "transport": s11.transport_text(),
                    "lml_rel_diff_vs_fused": abs(l11 - lf) / abs(lf),
                    "grad_rel_diff_vs_fused": float(np.abs(g11 - gf).max() / max(1e-300, np.abs(gf).max())),
                    "note": "ONE rank of the sharded sweep (nb = 512 block columns, rank-512 updates of K^-1 behind the "
                            "inverse, own tiles only) against `ms_per_step` of the fused sweep; per-rank compute times of a "
                            "2x4 grid: tools/sharded_replay.py, profiles/r05_sharded_replay*.json"}
                s11.close()
            except Exception as e:  # noqa: BLE001
                out["sharded_1x1"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_produce and g is not None:
            # secondary metric: Produce throughput at the same N, M = 1024 fresh
            # test points per call, host Z in / host mu, sigma out -- outside the timed region
            M = 1024 if N >= 1024 else 16

            def time_produce(m, reps=3):
                """One Produce call of m fresh test points (host Z in, host mu / sigma out), profiling events on
                its tile-kernel launches: wall time per call, union of the launch intervals, launched flops."""
                Zm = wl.test_points(m)
                g.Produce(Zm)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(reps):
                    g.Produce(Zm)
                t = (time.perf_counter() - t) / reps  # wall time per call, no events on the launches
                g.profile_enable(True)
                for _ in range(reps):
                    g.Produce(Zm)
                cross_ms, cross_n = g.profile_read_aux(2)
                _, launches, flops, busy_ms = g.profile_read()
                g.profile_enable(False)
                # algorithmic work of the solve (gp/gp.go:337-342 as ONE triangular solve): N^2 M flop
                alg = float(N) * N * m
                if m <= (64 if prec == 64 else 16):  # (float factors: the one-pass substitution up to 16 test points)
                    # few test points: ONE persistent launch that reads the factor once (trsm_small.hip) -- bound by
                    # the pass over the lower triangle, 8 N^2 / 2 bytes (bytes of the HBM-bound sub-steps)
                    tri = esz * float(N) * N / 2.0
                    d = {"m": m, "ms_per_call": t * 1e3, "test_points_per_s": m / t,
                         "roofline": {"bound": "hbm", "achieved": tri / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": tri / t / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_call": tri,
                                      "algorithmic_flops_per_call": alg,
                                      "note": "one pass over the factor's lower triangle / wall time of the whole call "
                                              "(Kstar, mean, substitution, norms, host copies); the substitution is a "
                                              "chain of N / 256 dependent block steps inside one launch: latency-, not "
                                              "bandwidth-bound at this N"}}
                else:
                    d = {"m": m, "ms_per_call": t * 1e3, "test_points_per_s": m / t,
                         "roofline": {"bound": "mfma", "achieved": alg / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                                      "frac": alg / t / 1e12 / peak,
                                      "algorithmic_flops_per_call": alg,
                                      "kernel_busy_ms_per_call": busy_ms / reps,
                                      "frac_while_kernel_runs": (alg / (busy_ms / reps * 1e-3) / 1e12 / peak) if busy_ms else None,
                                      "launches_per_call": launches / reps, "launched_flops_per_call": flops / reps}}
                if cross_n:
                    cb = esz * N * m
                    d["cross_kernel"] = {"algorithmic_bytes": cb, "ms": cross_ms / cross_n,
                                         "GBps": cb / (cross_ms / cross_n * 1e-3) / 1e9}
                return d

            out["produce"] = time_produce(M)
            out["produce"]["note"] = ("Kstar build + mu = Kstar^T alpha + blocked solve V^T = Kstar^T L^-T (N^2 M flop on the "
                                      "tile kernel, the test points' tile rows on up to 4 independent chains) + column "
                                      "norms, factor resident; roofline.frac = N^2 M / wall time of the whole call")
            if N >= 4096:
                out["produce"]["m_sweep"] = [time_produce(m, reps=3) for m in (1, 16, 64, 8192)]
                # the reference's harness order (tutorial/tutorial.go:170-179): Observe, then Produce of ONE point right
                # behind it -- the gradient preparation of the eager Observe still occupies the GPU
                Z1 = wl.test_points(1)
                g.Observe(wl.log_theta(1)); g.Produce(Z1)
                torch.cuda.synchronize()
                g.Observe(wl.log_theta(2))
                tb = time.perf_counter()
                g.Produce(Z1)
                tb = time.perf_counter() - tb
                torch.cuda.synchronize()
                out["produce"]["one_point_right_behind_observe_ms"] = tb * 1e3
        out_holder["line"] = out
        out_holder["measurement_done"] = True  # on every rank: the replica value needs nothing collective any more

    # ---- N > 1, replica configs: the transport's pre-flight, then ONE evaluation sharded over all ranks --
    sharded = None
    if world > 1 and not sharded_value:
        pw.soft = True
        try:
            preflight = run_preflight()
        except Exception as e:  # noqa: BLE001
            preflight = {"error": repr(e)[:300]}
        if out is not None:
            out["preflight"] = preflight
            out["transport_ok"] = "error" not in preflight
            out["rccl_ranks"] = int(preflight["comm_ranks"]) if preflight.get("rccl") else 0
    if world > 1 and not sharded_value and not args.no_sharded and "error" not in preflight:
        try:
            from gogp_amd.sharded import ShardedGP
            sg = ShardedGP(D, simil, noise, X=X, Y=y, device=local_rank)
            sg.Observe(wl.log_theta(0))
            sg.Gradient()
            sync()
            t0 = time.perf_counter()
            nrep = 3
            for k in range(nrep):
                lml_s = sg.Observe(wl.log_theta(1 + k))
                grad_s = sg.Gradient()
            sync()
            dts = gd.max_over_ranks((time.perf_counter() - t0) / nrep, device=red_dev)
            # same theta on a single GPU for the agreement check
            lml_1 = g.Observe(wl.log_theta(nrep))
            grad_1 = g.Gradient()
            sharded = {"ms_per_eval": dts * 1e3, "n_gpus": world, "evals_per_s": 1.0 / dts,
                       "layout": "2-D block-cyclic, grid %s, %s" % (sg.grid_text(), sg.transport_text()),
                       "bytes_per_rank": sg.local_bytes(),
                       "lml_rel_diff_vs_single_gpu": abs(lml_s - lml_1) / abs(lml_1),
                       "grad_rel_diff_vs_single_gpu":
                           float(np.abs(grad_s - grad_1).max() / max(1.0, np.abs(grad_1).max()))}
            sg.close()
        except Exception as e:  # noqa: BLE001
            sharded = {"error": repr(e)[:300]}
    elif sharded_value:
        # agreement of the sharded evaluation with ONE GPU running the same evaluation alone
        try:
            kk = warmup + steps - 1
            if rank == 0:
                g1 = G.GP(D, simil, noise, X=X, Y=y, device=local_rank, precision=prec)
                lml_1 = g1.Observe(wl.log_theta(kk))
                grad_1 = g1.Gradient()
                g1.close()
                sharded = {"bytes_per_rank": sg.local_bytes(),
                           "lml_rel_diff_vs_single_gpu": abs(lml - lml_1) / abs(lml_1),
                           "grad_rel_diff_vs_single_gpu":
                               float(np.abs(grad - grad_1).max() / max(1.0, np.abs(grad_1).max()))}
            sync()
            sg.close()
        except Exception as e:  # noqa: BLE001
            sharded = {"error": repr(e)[:300]}
    if wd is not None:
        wd.cancel()

    if rank == 0:
        if lbfgs_sharded is not None:
            out["lbfgs_loop"] = lbfgs_sharded
        if sharded is not None:
            out["sharded_evaluation"] = sharded
            if "error" in sharded and world > 1:
                out["transport_ok"] = False  # the transport came up but the sharded evaluation over it failed
        if world == 1 and not args.no_cpu_baseline:
            def gpu_fn(Xs, ys, x, Z):
                if len(ys) == N and g is not None:
                    g2 = g  # the full workload is already resident
                else:
                    g2 = G.GP(D, simil, noise, X=Xs, Y=ys, device=local_rank, precision=prec)
                v = g2.Observe(x)
                gr = g2.Gradient()
                mu, sigma = g2.Produce(Z)
                if g2 is not g:
                    g2.close()
                return v, gr, mu, sigma
            full = N <= ORACLE_FULL_N_LIMIT
            sample = N if full else min(args.cpu_sample_n, N)
            cb, errs = cpu_baseline(wl, sample, gpu_fn,
                                    second_sample_n=(8192 if full and N > 8192 else 0))
            out["cpu_baseline"] = cb
            out["parity_vs_oracle"] = errs
            out["lml_rel_err_vs_oracle"] = errs["lml_rel_err_vs_oracle"]
            out["grad_rel_err_vs_oracle"] = errs["grad_rel_err_vs_oracle"]
            out["mu_rel_err_vs_oracle"] = errs["mu_rel_err_vs_oracle"]
            out["sigma_rel_err_vs_oracle"] = errs["sigma_rel_err_vs_oracle"]
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
