"""Host enqueue time of one factorisation against the time until its LML is known (option debug_enqueue), per option set.
usage: python3 tools/enqueue_probe.py N D "opt=val ..." ["opt=val ..." ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G, kernel, synth
N, D = int(sys.argv[1]), int(sys.argv[2])
X, y = synth.make_inputs(N, D, 20251114 + 2)
x = np.log([1.0, np.sqrt(D / 6.0), 0.1])
for grp in sys.argv[3:]:
    g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
    for o in grp.split():
        if o != "-":
            g.set_option(o.split("=")[0], int(o.split("=")[1]))
    g.Observe(x); g.Gradient()
    print("==", grp, flush=True)
    g.set_option("debug_enqueue", 1)
    for k in range(3):
        g.Observe(x + 0.01 * k); g.Gradient()
    g.close()
