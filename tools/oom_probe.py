import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gogp_amd import kernel
from gogp_amd import gp as G
rng = np.random.default_rng(0)
g = G.GP(2, kernel.Scaled(kernel.Normal), kernel.UniformNoise)
n = 140000
g.X, g.Y = rng.uniform(0, 1, (n, 2)), rng.normal(size=n)
try:
    g.Observe(np.log([1.0, 0.5, 0.1]))
    print("UNEXPECTED: no error")
except G.GogpError as e:
    print("error as expected:", e)
# the same handle must still work at a small size
n = 500
g.X, g.Y = rng.uniform(0, 1, (n, 2)), rng.normal(size=n)
print("small problem after the failure: lml = %.6f" % g.Observe(np.log([1.0, 0.5, 0.1])), "grad", g.Gradient())
