import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import kernel
from gogp_amd import gp as G
rng = np.random.default_rng(0)
def make(n, seed):
    r = np.random.default_rng(seed)
    X = r.uniform(0, 1, (n, 3)); y = np.sin(6 * X).sum(1) + 0.1 * r.normal(size=n)
    return X, (y - y.mean()) / y.std()
cases = [(1500, 1), (2100, 2), (900, 3), (1800, 4)]
xs = [np.log([1.0, 0.5, 0.2]) + 0.01 * k for k in range(12)]
def run(case, out):
    X, y = make(*case)
    g = G.GP(3, kernel.Scaled(kernel.Matern32), kernel.UniformNoise, X=X, Y=y)
    res = []
    for x in xs:
        res.append((g.Observe(x), g.Gradient().copy(), g.Produce(X[:5])[0].copy()))
    g.close(); out.append(res)
seq = []
for c in cases:
    run(c, seq)
par = [[] for _ in cases]
ths = [threading.Thread(target=run, args=(c, par[i])) for i, c in enumerate(cases)]
[t.start() for t in ths]; [t.join() for t in ths]
ok = True
for i in range(len(cases)):
    for a, b in zip(seq[i], par[i][0]):
        ok &= (a[0] == b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
print("4 handles in 4 threads, 12 evaluations each: bitwise equal to sequential:", ok)
