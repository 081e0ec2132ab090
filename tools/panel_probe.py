"""One 128-column step of the Cholesky chain (panel128.hip) alone on the GPU: HIP-event time per launch for several panel
heights, the in-kernel phase stamps of workgroup 0 (s_memtime: shader cycles, printed in units of 100), the error against numpy.
usage: python3 tools/panel_probe.py [rows_below,rows_below,...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import _lib

H = _lib.hooks()
rows_list = [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,64,1024,4096,16256".split(","))]
rng = np.random.default_rng(5)
for rb in rows_list:
    n = 128 + rb
    B = rng.normal(size=(n, 160))
    K = B @ B.T / 160 + np.eye(n) * 0.5
    A = np.ascontiguousarray(K[:, :128])
    L = np.zeros_like(A)
    st = (ctypes.c_uint64 * 72)()
    us = ctypes.c_double()
    rc = H.gogp_test_panel128(0, A.ctypes.data_as(_lib._dp), L.ctypes.data_as(_lib._dp), rb, 50, st, ctypes.byref(us))
    assert rc == 0, rc
    Ld = np.linalg.cholesky(K[:128, :128])
    ref = np.vstack([Ld, np.linalg.solve(Ld, K[128:, :128].T).T])
    err = np.abs(np.tril(L[:128]) - Ld).max(), (np.abs(L[128:] - ref[128:]).max() if rb else 0.0)
    s = np.array(list(st), dtype=np.int64)
    t0 = s[0]
    print("rows_below %5d: %.2f us per launch (HIP events, %d workgroups); error: diagonal block %.2e, panel rows %.2e" %
          (rb, us.value, max(1, rb // 64), err[0], err[1]))
    tick = 0.01  # hundreds of cycles
    print("   [x100 cycles] load %.2f, first pivot block %.2f (+ barrier %.2f)" % ((s[1] - t0) * tick, (s[2] - s[1]) * tick, (s[3] - s[2]) * tick))
    prev = s[3]
    for kb in range(7):
        a, b, c, d, e = s[8 + kb * 8: 8 + kb * 8 + 5]
        print("   step %d: phase A %.2f (+ barrier %.2f), pivot wave %.2f, last trailing wave done at %.2f, barrier released %.2f  = %.2f" %
              (kb, (a - prev) * tick, (b - a) * tick, (c - b) * tick, (d - b) * tick, (e - b) * tick, (e - prev) * tick))
        prev = e
    print("   last pivot block stored at %.2f after entry" % ((s[4] - t0) * tick))
# multi-slab workgroups (launches with more workgroups than compute units): time per launch against slabs per workgroup
for rb in (4096, 16256, 31744):
    n = 128 + rb
    A = np.ascontiguousarray(rng.normal(size=(n, 128)) * 0.01)
    A[:128] += np.eye(128) * 2.0
    for slabs in (1, 2, 4):
        L = np.zeros_like(A)
        us = ctypes.c_double()
        assert H.gogp_test_panel128_slabs(0, A.ctypes.data_as(_lib._dp), L.ctypes.data_as(_lib._dp), rb, slabs, 20, ctypes.byref(us)) == 0
        print("rows_below %5d, %d slab(s) per workgroup (%d workgroups): %.2f us per launch" % (rb, slabs, -(-(rb // 64) // slabs), us.value))
