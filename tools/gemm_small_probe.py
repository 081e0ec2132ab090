import sys, ctypes
sys.path.insert(0, '/root/repo')
from gogp_amd import _lib
H = _lib.hooks()
for mode, mt, nt, K in ((1, 2, 2, 256), (0, 2, 2, 256), (1, 2, 2, 128), (0, 31, 1, 128), (1, 4, 4, 256), (1, 8, 8, 512), (0, 2, 2, 16)):
    ms = ctypes.c_double(); tf = ctypes.c_double()
    rc = H.gogp_bench_gemm(0, mode, mt, nt, K, 50, ctypes.byref(ms), ctypes.byref(tf))
    print("mode %d mt %d nt %d K %d: rc %d %.2f us per launch" % (mode, mt, nt, K, rc, ms.value * 1e3))
