"""Where one block step of the one-right-hand-side substitution kernel (trsm_small.hip: trsv_granule_kernel) spends its
time: s_memrealtime stamps (100 MHz) per workgroup -- 0 courier start, 1 last v_j arrived, 2 w published, 3 courier
released for w, 4 w_B arrived, 5 v published.  usage: python3 tools/trsv_stamps.py [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gogp_amd import gp as G, kernel, synth, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
D = 8
X, y = synth.make_inputs(N, D, 20251114 + 2)
g = G.GP(D, kernel.Scaled(kernel.Normal), kernel.UniformNoise, X=X, Y=y)
g.Observe(np.log([1.0, np.sqrt(D / 6.0), 0.1])); g.Gradient()
Z = np.random.default_rng(1).uniform(0, 1, (1, D))
g.Produce(Z)
nwg = (N + 255) // 256 * 4
st = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
lib = ctypes.CDLL(_lib.LIB_PATH)
ptr = ctypes.c_void_p.in_dll(lib, "_ZN4gogp11g_ts_stampsE")
ptr.value = st.data_ptr()
g.Produce(Z)
torch.cuda.synchronize()
ptr.value = None
s = st.cpu().numpy().reshape(nwg, 8).astype(np.float64) * 0.01  # us
t0 = s[:, 0].min()
s = s - t0
vpub = s[:, 5].reshape(-1, 4)          # per block: its four workgroups' publish times
blk = vpub.max(axis=1)
print("kernel: first courier start 0, last v published %.1f us; %d blocks, %.2f us per block" % (blk[-1], len(blk), np.diff(blk).mean()))
print("courier starts spread: %.1f us" % (s[:, 0].max()))
# per block B >= 1: v_{B-1} published (max over its workgroups) -> this block's workgroups see it -> publish w -> see w_B -> publish v
hop_v = s[4:, 1].reshape(-1, 4) - blk[:-1, None]
cmp_w = (s[:, 2] - s[:, 1])[4:].reshape(-1, 4)
wpub = s[:, 2].reshape(-1, 4)[1:]
hop_w = s[4:, 4].reshape(-1, 4) - wpub.max(axis=1)[:, None]
cmp_v = (s[:, 5] - s[:, 4])[4:].reshape(-1, 4)
rel = (s[:, 3] - s[:, 2])[4:].reshape(-1, 4)
for name, a in (("v hop (last publisher -> courier sees all 256)", hop_v), ("L phase of the last block + reduce + store w", cmp_w),
                ("barrier: w stored -> courier released", rel), ("w hop (last sibling's store -> courier sees all)", hop_w),
                ("barrier + Dinv phase + store v", cmp_v)):
    print("%-52s mean %.2f  median %.2f  p90 %.2f us" % (name, a.mean(), np.median(a), np.percentile(a, 90)))
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "trsv_stamps_%d.npy" % N), s)
