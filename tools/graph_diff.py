"""Where the chain graph (graph = 1) differs from the streams (graph = 0) and the DAG graph (graph = 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G, kernel, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 700
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
D = 2
X, y = synth.make_inputs(N, D, 7)
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
base = np.log([1.0, 0.6, 0.1])
xs = np.array([base + 0.02 * c for c in range(k)])
res = {}
for mode in (0, 1, 2, 3):
    g.set_option("graph", mode)
    outs = []
    for r in range(4):
        lmls, grads, st = g.observe_gradient_candidates(xs)
        outs.append((lmls.copy(), grads.copy()))
    print("mode", mode, "nodes", g.graph_info(), "lml", outs[-1][0], "grad0", outs[-1][1][0])
    print("   calls equal among themselves:", [bool(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1])) for o in outs])
    res.setdefault(mode, outs[-1])
for m in (1, 2, 3):
    print("mode %d vs 0: lml diff %s grad diff %s" % (m, res[m][0] - res[0][0], np.abs(res[m][1] - res[0][1]).max()))
