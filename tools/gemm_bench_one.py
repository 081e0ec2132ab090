import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gogp_amd import gp as G
mode, mt, nt, K = [int(v) for v in sys.argv[1:5]]
ms, tf = G.bench_gemm(mode, mt, nt, K, reps=3)
print("mode %d %dx%d K=%d: %.3f ms %.2f TF" % (mode, mt, nt, K, ms, tf))
