"""What a launch of the fp64 tile kernel costs beyond its k-loop: RECT launches of 512 .. 8192 tiles (1 .. 16 rounds of
the 512 resident workgroups) at several K, with and without the C-tile read (GOGP_BENCH_GEMM_BETA0), fitted as
time = a (per launch) + rounds * (b + c * K).
usage: python3 tools/gemm_fixedcost.py"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from gogp_amd import gp as G
    out = []
    for mt, nt in [(16, 32), (32, 32), (32, 64), (64, 64), (64, 128)]:
        for K in (256, 768, 2048):
            ms, tf = G.bench_gemm(0, mt, nt, K, reps=8)
            out.append((mt * nt, K, ms, tf))
    print(json.dumps(out))
    sys.exit(0)
import numpy as np
for beta0 in (0, 1):
    env = dict(os.environ)
    if beta0:
        env["GOGP_BENCH_GEMM_BETA0"] = "1"
    res = json.loads(subprocess.check_output([sys.executable, __file__, "child"], env=env).decode().strip().splitlines()[-1])
    print("beta = %d" % (0 if beta0 else 1))
    A, y = [], []
    for tiles, K, ms, tf in res:
        print("  tiles %5d K %5d  %8.4f ms  %6.2f TFLOP/s" % (tiles, K, ms, tf))
        r = tiles / 512.0
        A.append([1.0, r, r * K])
        y.append(ms * 1e3)
    (a, b, c), *_ = np.linalg.lstsq(np.array(A), np.array(y), rcond=None)
    print("  fit: %.1f us per launch + rounds * (%.2f us + K * %.4f us)" % (a, b, c))
