"""bench.py as the driver invokes it: `python3 bench.py --gpus N ...` must run N ranks.

CPU tests: the launcher logic alone (child launch before any GPU call, argument relay, exit-code relay,
refusal when WORLD_SIZE and --gpus disagree) with the rank body replaced by bench.py's own test double
(GOGP_BENCH_STUB=1: the ranks meet over gloo, no GPU, no library).
GPU test: the real rank body, two gloo ranks sharing the one GPU of the box (RCCL refuses duplicate
devices), exactly the command the driver would issue for N = 2.
No reference counterpart (the reference is a single process: gp/gp.go:165-213).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GOGP_BENCH_STUB",
                        "GOGP_BENCH_STUB_EXIT", "GOGP_BENCH_STUB_EXIT_RANK", "GOGP_DIST_BACKEND")}
    env.update(kw)
    return env


def _one_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line: %r" % stdout
    return json.loads(lines[0])


def test_gpus_n_starts_n_ranks_and_relays_the_arguments():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", "2", "--nobs", "512", "--steps", "2"],
                       env=_env(GOGP_BENCH_STUB="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2  # two processes met in the all-reduce
    assert d["gpus_arg"] == 2 and d["config"] == 2 and d["nobs"] == 512 and d["steps"] == 2


def test_one_gpu_runs_in_process():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--config", "1"], env=_env(GOGP_BENCH_STUB="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["config"] == 1


def test_a_failing_rank_fails_the_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"],
                       env=_env(GOGP_BENCH_STUB="1", GOGP_BENCH_STUB_EXIT="7", GOGP_BENCH_STUB_EXIT_RANK="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_world_size_and_gpus_must_agree():
    # a launcher started ONE rank but the command line says two GPUs: never a silent one-rank run
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"],
                       env=_env(GOGP_BENCH_STUB="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    assert r.stdout.strip() == ""
    assert "WORLD_SIZE=1" in r.stderr and "--gpus 2" in r.stderr
    # ... and the other way round (two ranks launched, --gpus 1 by default)
    r = subprocess.run([sys.executable, BENCH], env=_env(GOGP_BENCH_STUB="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2


def test_launcher_does_not_touch_the_gpu_before_the_child_starts():
    """The launching path of bench.py must not import torch / the library (a process that has initialised
    the GPU may not start another on this pool): everything between argument parsing and the child
    launch is standard-library code."""
    src = open(BENCH).read()
    body = src[src.index("def launch_ranks"):src.index("class PhaseWatchdog")]
    assert "import torch" not in body and "gogp_amd" not in body
    main = src[src.index("def main():"):]
    launch_at = main.index("sys.exit(launch_ranks(")
    assert "import torch" not in main[:launch_at] and "gogp_amd" not in main[:launch_at]


@pytest.mark.gpu
def test_bench_two_ranks_as_the_driver_runs_it():
    """`python3 bench.py --gpus 2 ...` end to end: launcher, pre-flight (communicator, ring, all-reduce),
    replica evaluations and ONE evaluation sharded over both ranks.  The two ranks share the box's one
    GPU, hence the gloo process group and the host-callback transport."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", "2", "--nobs", "512", "--steps", "2",
                        "--no-cpu-baseline"],
                       env=_env(GOGP_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2
    assert d["preflight"]["comm_ranks"] == 2 and d["preflight"]["rccl"] is False
    assert d["rccl_ranks"] == 0  # not an RCCL run: the field is only filled from ncclCommCount (0, never null)
    assert d["transport_ok"] is True  # the transport that was asked for (the gloo rehearsal) came up
    assert d["value"] > 0 and "error" not in d
    sh = d["sharded_evaluation"]
    assert "error" not in sh, sh
    assert sh["lml_rel_diff_vs_single_gpu"] < 1e-9 and sh["grad_rel_diff_vs_single_gpu"] < 1e-7


@pytest.mark.gpu
def test_bench_sharded_config_runs_the_optimiser_over_the_shards():
    """configs[4] ("LML+grad inside L-BFGS hyperparameter loop") at a rehearsal size on two gloo ranks:
    `value` is the sharded evaluation and the line carries the optimiser loop over the sharded handle."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--config", "5", "--nobs", "1024", "--ndim", "8",
                        "--steps", "2", "--warmup", "1"],
                       env=_env(GOGP_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dtype"] == "f32"
    lb = d["lbfgs_loop"]
    assert "error" not in lb, lb
    assert lb["evaluations"] >= 2 and lb["lml_end"] >= lb["lml_start"]
    assert d["sharded_evaluation"]["lml_rel_diff_vs_single_gpu"] < 1e-4


def test_phase_watchdog_names_rank_and_phase_and_exits_nonzero(tmp_path):
    """A pre-flight phase that does not return (a hung ncclCommInitRank, a send/recv ring that never
    completes) must end the process non-zero with the rank and the phase on stderr, and rank 0 still prints a
    JSON line carrying the error -- the call that hangs sits in C with the GIL released, here a sleep."""
    script = tmp_path / "wd.py"
    script.write_text(
        "import sys, time, json, os\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "out = os.fdopen(os.dup(1), 'w')\n"
        "wd = bench.PhaseWatchdog(int(sys.argv[1]), out, {'line': {'metric': 'm', 'value': None}})\n"
        "with wd.phase('quick phase', 5):\n"
        "    pass\n"
        "with wd.phase('grouped send/recv ring (rank -> rank+1)', 1):\n"
        "    time.sleep(30)\n"
        "print('not reached')\n" % ROOT)
    for rank in (0, 3):
        r = subprocess.run([sys.executable, str(script), str(rank)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 4
        assert "rank %d" % rank in r.stderr and "grouped send/recv ring" in r.stderr and "1 s" in r.stderr
        assert "not reached" not in r.stdout
        if rank == 0:
            d = _one_line(r.stdout)
            assert "grouped send/recv ring" in d["error"] and d["value"] is None
        else:
            assert r.stdout.strip() == ""


def test_a_hang_in_the_extra_sharded_evaluation_keeps_the_replica_measurement():
    """N > 1, replica configuration: the line's value is complete before the additional evaluation sharded
    over all ranks starts; if that one hangs, rank 0 prints the line with the extra marked and every rank
    exits 0.  When the sharded evaluation IS the value (configs 4, 5), or nothing was measured, exit 3."""
    import io
    sys.path.insert(0, ROOT)
    import bench
    buf = io.StringIO()
    line = {"metric": "m", "value": 27.5, "n_gpus": 2}
    assert bench.collective_timeout(line, False, 0, 300, buf) == 0
    d = json.loads(buf.getvalue())
    assert d["value"] == 27.5 and "timed out after 300 s" in d["sharded_evaluation"]["error"] and "error" not in d
    assert d["transport_ok"] is False
    buf = io.StringIO()
    # ranks != 0 never hold a line (main() sets it on rank 0 only): what they know is that the replica timing is done
    assert bench.collective_timeout(None, False, 1, 300, buf, measurement_done=True) == 0 and buf.getvalue() == ""
    assert bench.collective_timeout(None, False, 1, 300, buf, measurement_done=False) == 3 and buf.getvalue() == ""
    assert bench.collective_timeout(None, True, 1, 300, buf, measurement_done=True) == 3 and buf.getvalue() == ""
    buf = io.StringIO()
    assert bench.collective_timeout({"metric": "m", "value": 1.0}, True, 0, 300, buf) == 3
    assert "collective timed out" in json.loads(buf.getvalue())["error"]
    buf = io.StringIO()
    assert bench.collective_timeout(None, False, 0, 300, buf) == 3
    assert json.loads(buf.getvalue())["value"] is None


def test_phase_watchdog_after_a_complete_replica_measurement_reports_inside_the_line(tmp_path):
    """Replica configuration: the pre-flight of the sharded transport runs AFTER the measurement; a phase
    that does not return is then reported as `preflight.error` inside the complete line and the ranks exit 0."""
    script = tmp_path / "wd_soft.py"
    script.write_text(
        "import sys, time, json, os\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "out = os.fdopen(os.dup(1), 'w')\n"
        "rank = int(sys.argv[1])\n"
        "holder = {'line': {'metric': 'm', 'value': 27.5} if rank == 0 else None}\n"
        "wd = bench.PhaseWatchdog(rank, out, holder)\n"
        "wd.soft = True\n"
        "with wd.phase('communicator init (unique id broadcast + ncclCommInitRank)', 1):\n"
        "    time.sleep(30)\n"
        "print('not reached')\n" % ROOT)
    for rank in (0, 1):
        r = subprocess.run([sys.executable, str(script), str(rank)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0
        assert "rank %d" % rank in r.stderr and "communicator init" in r.stderr
        if rank == 0:
            d = _one_line(r.stdout)
            assert d["value"] == 27.5 and "communicator init" in d["preflight"]["error"] and "error" not in d
            assert d["transport_ok"] is False and d["rccl_ranks"] == 0  # never null: a driver reads one boolean
        else:
            assert r.stdout.strip() == ""
