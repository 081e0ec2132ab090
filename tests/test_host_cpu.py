"""CPU-side tests (no GPU needed): the C-ABI library loads and exports every
symbol include/gogp_hip.h declares, host logic (descriptors, synthetic inputs,
rank plumbing), and the product path fails loudly without a HIP device."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from gogp_amd import _lib, kernel, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "gogp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gogp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.gogp_version().decode().startswith("gogp_hip")
    # the product library exports none of the measurement hooks; they live in their own library
    hk = open(os.path.join(ROOT, "include", "gogp_testhooks.h")).read()
    hk = re.sub(r"/\*.*?\*/", "", hk, flags=re.S)
    hooks_declared = set(re.findall(r"\b(gogp_[a-z0-9_]+)\s*\(", hk))
    assert hooks_declared == {name for name, _, _ in _lib.HOOK_SYMBOLS}
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in hooks_declared:
        assert not hasattr(raw, name), name
        assert hasattr(_lib.hooks(), name)


def test_descriptor_struct_layout_matches_header(lib):
    # a descriptor built in Python must validate in C: catches field-order drift
    d = kernel.build_desc(3, kernel.Scaled(kernel.ARD(kernel.Normal, 3)), kernel.UniformNoise)
    assert lib.gogp_desc_check(ctypes.byref(d)) == _lib.GOGP_OK
    assert lib.gogp_desc_ntheta_noise(ctypes.byref(d)) == 1
    d.terms[0].len_idx = 2  # 2 + ndim(3) > ntheta(4)
    assert lib.gogp_desc_check(ctypes.byref(d)) == _lib.GOGP_EARG
    d2 = kernel.build_desc(1, kernel.Periodic, None)
    assert lib.gogp_desc_check(ctypes.byref(d2)) == _lib.GOGP_OK
    assert lib.gogp_desc_ntheta_noise(ctypes.byref(d2)) == 0
    assert abs(d2.noise_std - 1e-5) < 1e-20  # gp/gp.go:43-48 default
    d2.terms[0].period_mult = 0.0
    assert lib.gogp_desc_check(ctypes.byref(d2)) == _lib.GOGP_EARG
    assert ctypes.sizeof(kernel.CTerm) == 32
    assert ctypes.sizeof(kernel.CDesc) == 32 + 4 * 32


def test_kernel_composition_layouts():
    # tutorial/barebones/kernel/kernel.go:14-18: theta = [c, l]
    k = kernel.Scaled(kernel.Matern32)
    assert k.NTheta() == 2 and k.terms[0].scale_idx == 0 and k.terms[0].len_idx == 1
    # tutorial/hyperpriors/kernel/kernel.go:12-25: theta = [c1, c2, l1, l2, p]
    hp = kernel.Sum([kernel.Scaled(kernel.Matern52),
                     kernel.Scaled(kernel.PeriodScaled(kernel.Periodic, 10.0))],
                    order=[0, 2, 1, 3, 4])
    assert hp.NTheta() == 5
    t0, t1 = hp.terms
    assert (t0.scale_idx, t0.len_idx) == (0, 2)
    assert (t1.scale_idx, t1.len_idx, t1.period_idx, t1.period_mult) == (1, 3, 4, 10.0)
    x = [1.3, 0.4, 0.9, 1.1, 0.07, 0.2, 1.5]
    want = (x[0] * kernel.Matern52.Observe([x[2], x[5], x[6]])
            + x[1] * kernel.Periodic.Observe([x[3], 10 * x[4], x[5], x[6]]))
    assert abs(hp.Observe(x) - want) < 1e-15
    with pytest.raises(TypeError):
        kernel.build_desc(1, lambda x: 0.0, None)
    # noise kernels: kernel/noise.go
    assert kernel.ConstantNoise(0.1).Observe([0.0]) == pytest.approx(0.01)
    assert kernel.UniformNoise.Observe([0.3, 7.0]) == pytest.approx(0.09)
    assert kernel.ScaledNoise(0.01).Observe([2.0, 0.0]) == pytest.approx(0.04)


def test_synthetic_inputs_are_reproducible():
    X1, y1 = synth.make_inputs(1000, 8, 20251116)
    X2, y2 = synth.make_inputs(1000, 8, 20251116)
    assert np.array_equal(X1, X2) and np.array_equal(y1, y2)
    assert 0.0 <= X1.min() and X1.max() < 1.0
    assert abs(X1.mean() - 0.5) < 0.02 and abs(y1.std() - 1.0) < 1e-12
    # prefix property: a counter-based stream gives the same first rows for any n
    X3, _ = synth.make_inputs(10, 8, 20251116)
    assert np.array_equal(X3, X1[:10])
    ths = [tuple(synth.log_theta_cycle(8, k)) for k in range(6)]
    assert all(ths[k] != ths[k + 1] for k in range(5))  # theta changes every step


def test_no_gpu_fails_loudly(lib):
    """No CPU fallback: without a HIP device gogp_create returns GOGP_EHIP and the
    Python mirror raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gogp_amd import gp
    with pytest.raises(gp.GogpError) as ei:
        gp.GP(1, kernel.Normal)
    assert ei.value.code == _lib.GOGP_EHIP
    v = ctypes.c_double()
    assert _lib.hooks().gogp_mfma_f64_peak(0, 10, ctypes.byref(v), None, None) == _lib.GOGP_EHIP


def test_product_does_not_import_oracle():
    """The product path must not route through oracle/ (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gogp_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
                assert "libgogp_oracle" not in src, fn


_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from gogp_amd import dist as gd
rank, world = gd.init("gloo")
assert world == 2
gd.barrier()
assert gd.max_over_ranks(1.0 + rank) == 2.0
ncand, width = 5, 4
mine = gd.my_candidates(ncand, rank, world)
vals = np.array([[i, 10.0 * i, rank, -i] for i in mine], dtype=float).reshape(len(mine), width)
allv = gd.gather_results(mine, vals, ncand, width)
for i in range(ncand):
    assert allv[i, 0] == i and allv[i, 1] == 10.0 * i and allv[i, 2] == i %% world
gd.barrier()
dist.destroy_process_group()
open(os.path.join(%(out)r, "rank%%d.ok" %% rank), "w").write("ok")
"""


def test_rank_plumbing_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def _newest_profile(suffix):
    """The newest round's tracked profile file profiles/rNN_<suffix>."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    assert hits, suffix
    return hits[-1]


def test_bench_reports_pmc_traffic_only_for_the_build_it_was_measured_on():
    """profiles/rNN_pmc_traffic.json (the newest round's) carries the build id of the library it was measured on (gogp_version():
    a hash of the library's sources); bench.py copies the number into roofline.traffic for that build only and
    says why not otherwise -- a stale file must not look like a measurement of the current kernel."""
    import json
    import bench
    d = json.load(open(_newest_profile("pmc_traffic.json")))
    assert d["build"] and d["3"]["N"] == 16384 and d["3"]["bytes_per_launch"] > 1e8
    e, why = bench.pmc_traffic(3, d["build"])
    assert why is None and e["bytes_per_launch"] == d["3"]["bytes_per_launch"]
    e, why = bench.pmc_traffic(3, "0123456789ab")
    assert e is None and "stale" in why and d["build"] in why
    e, why = bench.pmc_traffic(1, d["build"])   # round 5 measured configs 2-5 at their own sizes; config 1 has no pass
    assert e is None and "no PMC pass" in why
    for c in ("2", "4", "5"):
        if c in d:
            e, why = bench.pmc_traffic(int(c), d["build"])
            assert why is None and e["bytes_per_launch"] == d[c]["bytes_per_launch"]
    # the version string of the built library ends in a 12-digit hex build id
    from gogp_amd import _lib
    v = _lib.lib().gogp_version().decode()
    bid = v.split("build ")[-1]
    assert len(bid) == 12 and all(c in "0123456789abcdef" for c in bid), v


def test_roofline_is_recomputable_from_the_tracked_trace():
    """profiles/rNN_c3_roofline.json must follow from profiles/rNN_c3_kernel_trace.csv (the raw start / end
    timestamps of every dispatch) by tools/roofline_from_profiles.py, and the bench line of the same
    profiling call from both within box-to-box spread: the numbers the judge reads are recomputable."""
    import json
    import subprocess
    import sys
    prof = os.path.join(ROOT, "profiles")
    trace = _newest_profile("c3_kernel_trace.csv")
    rnd = os.path.basename(trace)[:3]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_from_profiles.py"), "16384",
                        "dgemm_nt_kernel", "78.6", trace], capture_output=True, text=True, check=True)
    again = json.loads(r.stdout)
    kept = json.load(open(os.path.join(prof, rnd + "_c3_roofline.json")))
    for k in ("union_ms_per_evaluation", "sum_ms_per_evaluation", "launches_per_evaluation", "frac_union"):
        assert abs(again[k] - kept[k]) <= 1e-9 * abs(kept[k]), (k, again[k], kept[k])
    assert again["evaluations_in_trace"] == kept["evaluations_in_trace"] == 6
    # frac = N^3 / union of the launch intervals / peak, nothing else
    assert abs(kept["frac_union"] - 16384.0 ** 3 / (kept["union_ms_per_evaluation"] * 1e-3) / 78.6e12) < 1e-12
    # the bench line's live HIP-event measurement of the same quantity, on another box
    line = json.load(open(os.path.join(prof, rnd + "_bench_c3.json")))
    assert abs(line["roofline"]["frac"] - kept["frac_union"]) < 0.03
    assert line["roofline"]["frac_wall"] <= line["roofline"]["frac"]
    # the traffic number the line carries is the stamped one, for the line's own build
    traffic = json.load(open(os.path.join(prof, rnd + "_pmc_traffic.json")))
    assert line["library"].endswith("build " + traffic["build"])
    assert abs(line["roofline"]["traffic"] - traffic["3"]["bytes_per_launch"]) < 1.0


def test_tracked_bench_line_carries_the_contract_fields():
    """The newest tracked default-configuration line (profiles/rNN_bench_c3.json) is BASELINE.json's metric on its
    headline configuration with the two objects the measurement contract asks for, the Produce roofline and -- clearly
    apart from `value` -- the mixed-precision option."""
    import json
    line = json.load(open(_newest_profile("bench_c3.json")))
    assert line["metric"].startswith("GP.Observe+Gradient evals/sec (fp64) at N=16384 D=8")
    assert line["dtype"] == "f64" and line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["config"]["candidates_per_step"] == 1 and "configs[2]" in line["config"]["workload"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 78.6 and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["frac_wall"] <= r["frac"] < 1.0
    assert abs(r["frac_wall"] - 16384.0 ** 3 / (line["ms_per_step"] * 1e-3) / 78.6e12) < 1e-9
    assert r["traffic"] and r["traffic"] > 1e8
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert line["lml_rel_err_vs_oracle"] < 1e-6 and line["mu_rel_err_vs_oracle"] < 1e-6 and line["sigma_rel_err_vs_oracle"] < 1e-6
    p = line["produce"]
    assert p["m"] == 1024 and abs(p["roofline"]["frac"] - 16384.0 ** 2 * 1024 / (p["ms_per_call"] * 1e-3) / 78.6e12) < 1e-9
    ms = [q["m"] for q in p["m_sweep"]]
    assert ms[0] == 1 and 64 in ms and ms[-1] == 8192
    for q in p["m_sweep"]:   # few test points: one pass over the factor, priced against HBM (round 5); many: the fp64 roof
        rq = q["roofline"]
        assert rq["bound"] == ("hbm" if q["m"] <= 64 else "mfma")
        if q["m"] <= 64:
            assert abs(rq["frac"] - 8.0 * 16384.0 ** 2 / 2 / (q["ms_per_call"] * 1e-3) / 8e12) < 1e-9
    m = line["mixed_precision_gradient"]  # an option beside the value, never the value
    assert m["lml_identical_to_native"] is True and m["grad_rel_diff_vs_native"] < 1e-6
    assert m["evals_per_s"] > line["value"] and "not `value`" in m["note"]

