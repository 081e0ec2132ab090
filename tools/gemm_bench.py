"""Rate of the fp64 tile kernel alone (hook gogp_bench_gemm), shape by shape; with GOGP_BENCH_GEMM_LD0=1 in
the environment every operand row aliases row 0, i.e. all operand loads hit in the caches (the kernel with
memory latency taken out); with GOGP_BENCH_GEMM_F32=1 the fp32 tile kernel (K counted in floats).
usage: python3 tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gogp_amd import gp as G
f32 = bool(os.environ.get("GOGP_BENCH_GEMM_F32"))
print("peak: %.1f TF%s" % (G.mfma_f32_peak(50000) if f32 else G.mfma_f64_peak(50000), "   (operand rows aliased: cache-resident)" if os.environ.get("GOGP_BENCH_GEMM_LD0") else ""))
for name, mode, mt, nt, K in [
    ("RECT 64x64 K=256", 0, 64, 64, 256), ("RECT 64x64 K=512", 0, 64, 64, 512), ("RECT 64x64 K=768", 0, 64, 64, 768),
    ("RECT 64x64 K=1024", 0, 64, 64, 1024), ("RECT 64x64 K=4096", 0, 64, 64, 4096), ("RECT 64x64 K=16384", 0, 64, 64, 16384),
    ("RECT 24x24 K=768 (4-wave)", 0, 24, 24, 768), ("RECT 24x24 K=4096 (4-wave)", 0, 24, 24, 4096),
    ("LOWER 120 K=256", 1, 120, 120, 256), ("LOWER 120 K=512", 1, 120, 120, 512), ("LOWER 120 K=768", 1, 120, 120, 768),
    ("LOWER 64 K=256", 1, 64, 64, 256), ("LOWER 32 K=256", 1, 32, 32, 256),
    ("RECT 32x96 K=256", 0, 32, 96, 256), ("RECT 96x32 K=256", 0, 96, 32, 256),
    ("RECT 126x2 K=256", 0, 126, 2, 256), ("RECT 64x2 K=256", 0, 64, 2, 256), ("RECT 16x2 K=256", 0, 16, 2, 256),
    ("LAUUM 128", 2, 128, 128, 0),
]:
    if f32 and K % 32:
        continue
    ms, tf = G.bench_gemm(mode, mt, nt, K if K else 32, reps=5)
    print("%-28s %8.3f ms  %6.2f TFLOP/s" % (name, ms, tf), flush=True)
