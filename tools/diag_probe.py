import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gogp_amd import gp as G
rng = np.random.default_rng(0)
B = rng.normal(size=(256, 256))
A = B @ B.T + 256 * np.eye(256)
L, X, st, us = G.diag256_check(A)
Lr = np.linalg.cholesky(A)
print("max |L-Lref| = %.3e, max |X - inv(L)| = %.3e, elapsed %.1f us" % (
    np.abs(L - Lr).max(), np.abs(X - np.linalg.inv(Lr)).max(), us))
names = {0: "h0 start", 1: "h0 loaded", 2: "h0 potrf done", 3: "h0 Lout written", 4: "h0 inverted",
         5: "h0 Dinv written", 8: "h1 start(L10 done)", 9: "h1 schur loaded", 10: "h1 potrf done",
         11: "h1 Lout written", 12: "h1 inverted", 13: "h1 Dinv written", 16: "x10 start",
         17: "U done", 18: "end"}
t0 = int(st[0])
prev = t0
for k in sorted(names):
    t = int(st[k])
    print("%-22s +%7d cyc  (total %7d)" % (names[k], t - prev, t - t0))
    prev = t

# sub-stamps of potrf128 (half 0): st[20] = solve kb=0 done, st[21] = wave 0's update + potrf16(1)
# done, st[23] = solve kb=1 done, st[24] = potrf16(2) done
t1, s0, p1, s1, p2 = int(st[1]), int(st[20]), int(st[21]), int(st[21]), int(st[24])
print("potrf16(0)+solve(0): %d | update+potrf16(1): %d | barrier+solve(1): %d | update+potrf16(2): %d" % (
    s0 - t1, p1 - s0, s1 - p1, p2 - s1))
print("L10 product %d + epilogue %d | Schur product %d + epilogue %d" % (
    int(st[27]) - int(st[5]), int(st[8]) - int(st[27]), int(st[26]) - int(st[8]), int(st[9]) - int(st[26])))
