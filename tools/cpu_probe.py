"""The CPU baseline of bench.py alone, phase by phase: FastOracle (scipy/OpenBLAS potrf + potri + potrs,
C/OpenMP Gram and gradient pair loops) at several sizes in both orders, with the thread pools in effect.
usage: python3 tools/cpu_probe.py [cores] [import_torch]"""
import os, sys, time
sys.path.insert(0, '.')
cores = int(sys.argv[1]) if len(sys.argv) > 1 else 16
os.environ["OMP_NUM_THREADS"] = str(cores)
if len(sys.argv) > 2:
    import torch  # noqa: F401  (bench.py has torch loaded when the baseline runs)
import numpy as np
from threadpoolctl import threadpool_limits, threadpool_info
from gogp_amd import configs
from oracle.oracle import FastOracle
threadpool_limits(limits=cores)
print([(i['internal_api'], i['num_threads'], i.get('threading_layer'), os.path.basename(i['filepath'])) for i in threadpool_info()])
print("affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None)
wl = configs.workload(3)
X, y = wl.inputs()
for n in (1024, 4096, 8192, 16384, 8192, 4096):
    o = FastOracle(wl.D, wl.simil, wl.noise, block=2048)
    o.set_data(X[:n], y[:n])
    t0 = time.time(); lml = o.Observe(wl.log_theta(0)); t1 = time.time(); g = o.Gradient(); t2 = time.time()
    print(n, "observe %.2f s gradient %.2f s" % (t1 - t0, t2 - t1), {k: round(v, 3) for k, v in o.timings.items()}, flush=True)
