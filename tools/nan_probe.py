import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import kernel
from gogp_amd import gp as G
rng = np.random.default_rng(0)
x = np.log([1.0, 0.5, 0.1])
for n in (50, 700):
    X = rng.uniform(0, 1, (n, 2)); y = rng.normal(size=n)
    Xn = X.copy(); Xn[n // 2, 1] = np.nan
    g = G.GP(2, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=Xn, Y=y)
    try:
        print(n, "NaN in X: lml =", g.Observe(x))
    except G.GogpError as e:
        print(n, "NaN in X ->", type(e).__name__, e)
    yn = y.copy(); yn[3] = np.nan
    g.X, g.Y = X, yn
    try:
        lml = g.Observe(x); print(n, "NaN in y: lml =", lml, "grad =", g.Gradient())
    except G.GogpError as e:
        print(n, "NaN in y ->", type(e).__name__, e)
    g.X, g.Y = X, y
    print(n, "clean again: lml = %.6f" % g.Observe(x))
    g.close()
