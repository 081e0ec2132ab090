/*
 * gogp_testhooks.h -- measurement and diagnostic entry points of
 * gogp_amd/libgogp_testhooks.so (tests/, tools/ and bench.py's roofline calibration).
 *
 * NOT part of the drop-in boundary: the product library libgogp_hip.so
 * (include/gogp_hip.h) does not export these.  The hook library links the product
 * library and drives its internal launchers.
 */
#ifndef GOGP_TESTHOOKS_H
#define GOGP_TESTHOOKS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Micro-benchmark used to calibrate the fp64 MFMA roofline: every SIMD issues
 * `iters` x 8 back-to-back v_mfma_f64_16x16x4_f64 from two waves; returns the
 * achieved TFLOP/s and (optionally) the shader cycles per MFMA on one SIMD and
 * the shader clock in MHz observed by one wave during the run. */
int gogp_mfma_f64_peak(int device, int iters, double *tflops, double *cyc_per_mfma,
                       double *clock_mhz);
/* The same for the fp32 path's instruction, v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s spec). */
int gogp_mfma_f32_peak(int device, int iters, double *tflops, double *cyc_per_mfma,
                       double *clock_mhz);

/* Stand-alone fp64 GEMM test hook: C(MxN,row-major) = beta*C + alpha*A(MxK)*B(NxK)^T
 * on host buffers (copied to the device and back); M,N multiples of 128,
 * K multiple of 16.  Exists so the tile kernel can be parity-tested in
 * isolation against a host reference. */
int gogp_test_dgemm_nt(int device, int64_t M, int64_t N, int64_t K, double alpha,
                       const double *A, const double *B, double beta, double *C);

/* Benchmark hook for the tile kernel: `reps` launches of one shape (mode 0 RECT
 * mt x nt tiles, 1 LOWER mt x mt, 2 LAUUM mt x mt with K = mt*128) on device
 * buffers; returns ms per launch and TFLOP/s on the flops launched. */
int gogp_bench_gemm(int device, int mode, int mt, int nt, int64_t K, int reps,
                    double *ms_per_launch, double *tflops);

/* Diagnostic hook for the diagonal-block kernel: factor + invert one 256x256 SPD
 * block given on the host (row-major, lower triangle used); returns the factor,
 * its dense inverse, 24 in-kernel s_memtime stamps of a diagnostic build and the
 * HIP-event time (us) of the product build. */
int gogp_test_diag256(int device, const double *A, double *Lout, double *Dinv,
                      unsigned long long *stamps, double *elapsed_us);

/* Cycles per instruction (s_memtime) of one wave alone on its SIMD, the operations of the pivot chain (pivot16.h):
 * out8[0..7] = dependent v_fma_f64, independent v_fma_f64, dependent v_mov_b64_dpp row_newbcast, independent v_mov_b64_dpp,
 * dependent v_rsq_f64, independent v_rsq_f64, dependent v_mul_f64, one dependent (dpp, fma) pair. */
int gogp_test_valu_cost(int device, double *out8);

/* Diagnostic hook for one 128-column step of the Cholesky chain (panel128.hip): a (128 + rows_below) x 128 panel given on
 * the host (row-major, ld 128, lower triangle of the diagonal block used; rows_below a multiple of 64); returns the factor
 * of the diagonal block with the solved rows under it (same layout), 72 in-kernel s_memtime stamps of a diagnostic build
 * and the HIP-event time (us per launch over `reps` launches) of the product build. */
int gogp_test_panel128(int device, const double *A, double *Lout, int64_t rows_below, int reps,
                       unsigned long long *stamps, double *elapsed_us);

/* The product build of that step with the number of 64-row slabs per workgroup forced (0: by size): the factor and the
 * solved rows (same layout) and the HIP-event time per launch.  The result must not depend on `slabs`. */
int gogp_test_panel128_slabs(int device, const double *A, double *Lout, int64_t rows_below, int slabs, int reps,
                             double *elapsed_us);

/* Per-rank replay of a sharded evaluation (tools/sharded_replay.py): makes handle `h` (a
 * gogp_handle of the product library) rank `rank` of a prow x pcol grid ALONE on its GPU -- nothing is sent, a receive
 * zero-fills its buffer, an all-reduce is the identity.  Every launch of that rank's share of the sweep runs with its
 * real shape, so the wall time of Observe + Gradient is the rank's compute time; the values returned mean nothing. */
int gogp_test_dist_init_replay(void *h, int rank, int nranks, int prow, int pcol);

#ifdef __cplusplus
}
#endif
#endif /* GOGP_TESTHOOKS_H */
