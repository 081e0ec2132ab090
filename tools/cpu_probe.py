import os, sys, time
sys.path.insert(0, '.')
cores = int(sys.argv[1]) if len(sys.argv) > 1 else 16
os.environ["OMP_NUM_THREADS"] = str(cores)
import numpy as np
from threadpoolctl import threadpool_limits, threadpool_info
from gogp_amd import configs
from oracle.oracle import FastOracle
threadpool_limits(limits=cores)
print([ (i['internal_api'], i['num_threads']) for i in threadpool_info()])
wl = configs.workload(3)
X, y = wl.inputs()
for n in (1024, 4096, 8192, 16384, 8192, 4096):
    o = FastOracle(wl.D, wl.simil, wl.noise, block=2048)
    o.set_data(X[:n], y[:n])
    t0 = time.time(); lml = o.Observe(wl.log_theta(0)); t1 = time.time(); g = o.Gradient(); t2 = time.time()
    print(n, "observe %.2f s gradient %.2f s" % (t1 - t0, t2 - t1), flush=True)
