"""The fp32 path's gradient on the ill-conditioned golden case, component by component, against the oracle --
with the options that could matter (refinement steps of alpha, eager / lazy inverse, closed forms on / off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gogp_amd import gp as G, kernel
from oracle.oracle import FastOracle
d = np.load(os.path.join(ROOT, "tests", "golden", "fp32_illcond_matern32.npz"))
X, y, x = d["X"], d["y"], d["x"]
simil, noise = kernel.Scaled(kernel.Matern32), kernel.ScaledNoise(0.01)  # tests/cases.py: "matern32"
o = FastOracle(2, simil, noise)
o.set_data(X, y)
lml_o, grad_o = o.Observe(x), o.Gradient()
scale = np.abs(grad_o).max()
print("theta", np.exp(x), "oracle lml %.9f grad %s" % (lml_o, grad_o))
for prec, opts in ((32, {}), (32, {"diag_fp64": 0}), (32, {"refine_steps": 3}), (32, {"eager": 0}), (32, {"trace_fp64": 0}), (64, {"gradient_precision": 32}), (64, {})):
    g = G.GP(2, simil, noise, X=X, Y=y, precision=prec)
    for k, v in opts.items():
        g.set_option(k, v)
    lml = g.Observe(x)
    grad = g.Gradient()
    a = g.Alpha
    print("precision %d %-28s lml rel %.2e  grad err / scale %s  |alpha - oracle| rel %.2e" %
          (prec, opts, abs(lml - lml_o) / abs(lml_o), np.abs(grad - grad_o) / scale,
           np.abs(a - o.alpha).max() / np.abs(o.alpha).max() if hasattr(o, "alpha") else float("nan")))
    g.close()
