"""Cycles per instruction of the pivot chain's operations (one wave alone on its SIMD): gogp_test_valu_cost."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import _lib
out = np.zeros(8)
assert _lib.hooks().gogp_test_valu_cost(0, out.ctypes.data_as(_lib._dp)) == 0
for name, v in zip(["v_fma_f64 dependent", "v_fma_f64 independent", "v_mov_b64_dpp dependent (+ s_nop 1)", "v_mov_b64_dpp independent",
                    "v_rsq_f64 dependent", "v_rsq_f64 independent", "v_mul_f64 dependent", "dpp + fma dependent pair"], out):
    print("%-40s %6.1f cycles" % (name, v))
