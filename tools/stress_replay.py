"""Replays the case tools/stress.py saved on a mismatch (gpurun_out/stress_mismatch.npz) on fresh handles:
fp64, fp32, fp32 with the lazy inverse, and prints the gradient components beside the oracle's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cases import CASES
from gogp_amd import gp as G
from oracle.oracle import FastOracle
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stress_mismatch.npz", allow_pickle=True)
name = str(d["name"])
case = [c for c in CASES if c[0] == name][0]
_, D, simil, noise, ts, tn = case
X, y, x = d["X"], d["y"], d["x"]
o = FastOracle(D, simil, noise); o.set_data(X, y); lo = o.Observe(x); go = o.Gradient()
print("case", name, "n", len(y), "D", D, "x", x, "saved opts", d["opts"], "order", d["order"])
print("oracle        lml %.9f grad %s" % (lo, go))
for prec, opts in ((64, {}), (32, {}), (32, {"eager": 0}), (32, {"eager": 0, "refine_steps": 2})):
    for rep in range(2):
        g = G.GP(D, simil, noise, precision=prec)
        for k, v in opts.items():
            g.set_option(k, v)
        g.X, g.Y = X, y
        l = g.Observe(x); gr = g.Gradient()
        if rep: gr = g.Gradient()
        print("prec %d %-28s rep %d lml %.9f grad %s  rel err %.3e" % (
            prec, opts, rep, l, gr, np.abs(gr - go).max() / max(1.0, np.abs(go).max())))
        g.close()
