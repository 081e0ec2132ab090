"""The 2-D sharded code path on ONE rank (1x1 grid): time per evaluation (profile it with rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import configs
from gogp_amd.sharded import ShardedGP
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = configs.workload(cfg)
X, y = wl.inputs()
sh = ShardedGP(wl.D, wl.simil, wl.noise, X=X, Y=y, device=0, transport="callbacks")
sh.Observe(wl.log_theta(0)); sh.Gradient()
t0 = time.perf_counter()
for k in range(3):
    lml = sh.Observe(wl.log_theta(1 + k)); gr = sh.Gradient()
print("config %d sharded 1x1: %.1f ms/eval" % (cfg, (time.perf_counter() - t0) / 3 * 1e3))
sh.close()
