// testhooks.hip -- measurement / diagnostic entry points (include/gogp_testhooks.h).
// Built into gogp_amd/libgogp_testhooks.so, which links libgogp_hip.so and calls its
// internal launchers; none of this is part of the product ABI (include/gogp_hip.h).
#include <cstdlib>
#include <stdio.h>

#include "common.h"
#include "../../include/gogp_testhooks.h"

struct gogp_handle;
namespace gogp {
int dist_init_replay(gogp_handle *h, int rank, int nranks, int prow, int pcol);  // dist2d.hip
}
extern "C" int gogp_test_dist_init_replay(void *h, int rank, int nranks, int prow, int pcol) {
  return gogp::dist_init_replay(static_cast<gogp_handle *>(h), rank, nranks, prow, pcol);
}

namespace gogp_th {
void launch_diag256(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl,
                    double *Dinv, int64_t row0, int64_t nvalid, long long *info);
void launch_diag256_stamped(hipStream_t s, const double *A, double *Lout, double *Dinv,
                            long long *info, unsigned long long *stamps);
void launch_panel128_stamped(hipStream_t s, const double *A, double *Lout, int64_t rows_below, long long *info,
                             unsigned long long *stamps);
}

namespace gogp {
typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---- fp64 MFMA issue-rate microbenchmark (roofline calibration) -------------
// 8 independent accumulators held in AGPRs by inline asm (the builtin form makes
// hipcc shuttle loop-carried accumulators between VGPRs and AGPRs every
// iteration, which under-reads the rate).  Wave 0 of block 0 also reports shader
// cycles (s_memtime) and wall ticks (s_memrealtime, 100 MHz) around its loop.
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(int iters, double *sink,
                                                            unsigned long long *clk) {
  f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  int cnt = iters;
  // the whole loop lives in one asm statement so that the accumulators stay in
  // AGPRs across iterations
  asm volatile(
      "1:\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %9, %10, %0\n\t"
      "v_mfma_f64_16x16x4_f64 %1, %9, %10, %1\n\t"
      "v_mfma_f64_16x16x4_f64 %2, %9, %10, %2\n\t"
      "v_mfma_f64_16x16x4_f64 %3, %9, %10, %3\n\t"
      "v_mfma_f64_16x16x4_f64 %4, %9, %10, %4\n\t"
      "v_mfma_f64_16x16x4_f64 %5, %9, %10, %5\n\t"
      "v_mfma_f64_16x16x4_f64 %6, %9, %10, %6\n\t"
      "v_mfma_f64_16x16x4_f64 %7, %9, %10, %7\n\t"
      "s_sub_u32 %8, %8, 1\n\t"
      "s_cmp_lg_u32 %8, 0\n\t"
      "s_cbranch_scc1 1b"
      : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7),
        "+s"(cnt)
      : "v"(a), "v"(b)
      : "scc");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  if (s == 12345.678) sink[0] = s;  // keep the chains live
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

// the fp32 twin: v_mfma_f32_32x32x2_f32 (4096 flop, 16 passes), eight independent accumulators
typedef float f32x16p __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_f32_peak_kernel(int iters, float *sink, unsigned long long *clk) {
  f32x16p c0 = {0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  float a = 1.0f + 1e-6f * threadIdx.x, b = 1.0f - 1e-6f * threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  int cnt = iters;
  asm volatile(
      "1:\n\t"
      "v_mfma_f32_32x32x2_f32 %0, %9, %10, %0\n\t"
      "v_mfma_f32_32x32x2_f32 %1, %9, %10, %1\n\t"
      "v_mfma_f32_32x32x2_f32 %2, %9, %10, %2\n\t"
      "v_mfma_f32_32x32x2_f32 %3, %9, %10, %3\n\t"
      "v_mfma_f32_32x32x2_f32 %4, %9, %10, %4\n\t"
      "v_mfma_f32_32x32x2_f32 %5, %9, %10, %5\n\t"
      "v_mfma_f32_32x32x2_f32 %6, %9, %10, %6\n\t"
      "v_mfma_f32_32x32x2_f32 %7, %9, %10, %7\n\t"
      "s_sub_u32 %8, %8, 1\n\t"
      "s_cmp_lg_u32 %8, 0\n\t"
      "s_cbranch_scc1 1b"
      : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7), "+s"(cnt)
      : "v"(a), "v"(b)
      : "scc");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + c4[4] + c5[5] + c6[6] + c7[7];
  if (s == 12345.678f) sink[0] = s;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

int mfma_f32_peak(int iters, double *tflops, double *cyc_per_mfma, double *clock_mhz) {
  float *sink = nullptr;
  unsigned long long *clk = nullptr;
  if (hipMalloc(&sink, 8) != hipSuccess) return GOGP_EHIP;
  if (hipMalloc(&clk, 16) != hipSuccess) return GOGP_EHIP;
  hipDeviceProp_t prop;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return GOGP_EHIP;
  const int blocks = prop.multiProcessorCount * 2;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_f32_peak_kernel, dim3(blocks), dim3(256), 0, 0, iters / 4 + 1, sink,
                     (unsigned long long *)nullptr);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_f32_peak_kernel, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
  (void)hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return GOGP_EHIP;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 32 * 32 * 2;
  *tflops = flops / (ms * 1e-3) / 1e12;
  unsigned long long h[2] = {0, 0};
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  if (cyc_per_mfma) *cyc_per_mfma = (double)h[0] / ((double)iters * 8.0 * 2.0);
  if (clock_mhz) *clock_mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  (void)hipFree(clk);
  return GOGP_OK;
}

// tflops: achieved rate with every SIMD issuing; cyc_per_mfma / clock_mhz from wave 0.
int mfma_f64_peak(int iters, double *tflops, double *cyc_per_mfma, double *clock_mhz) {
  double *sink = nullptr;
  unsigned long long *clk = nullptr;
  if (hipMalloc(&sink, 8) != hipSuccess) return GOGP_EHIP;
  if (hipMalloc(&clk, 16) != hipSuccess) return GOGP_EHIP;
  hipDeviceProp_t prop;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return GOGP_EHIP;
  const int blocks = prop.multiProcessorCount * 2;  // 8 waves per CU = 2 per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, iters / 4 + 1, sink,
                     (unsigned long long *)nullptr);  // warm-up (clock ramp)
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, iters, sink, clk);
  (void)hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return GOGP_EHIP;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 16 * 16 * 4;
  *tflops = flops / (ms * 1e-3) / 1e12;
  unsigned long long h[2] = {0, 0};
  (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  // two waves share a SIMD: cycles per MFMA issued on that SIMD
  if (cyc_per_mfma) *cyc_per_mfma = (double)h[0] / ((double)iters * 8.0 * 2.0);
  if (clock_mhz) *clock_mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  (void)hipFree(clk);
  return GOGP_OK;
}


}  // namespace gogp

using namespace gogp;

extern "C" int gogp_mfma_f64_peak(int device, int iters, double *tflops, double *cyc_per_mfma,
                                  double *clock_mhz) {
  if (!tflops || iters <= 0) return GOGP_EARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  return mfma_f64_peak(iters, tflops, cyc_per_mfma, clock_mhz);
}

extern "C" int gogp_mfma_f32_peak(int device, int iters, double *tflops, double *cyc_per_mfma,
                                  double *clock_mhz) {
  if (!tflops || iters <= 0) return GOGP_EARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  return mfma_f32_peak(iters, tflops, cyc_per_mfma, clock_mhz);
}

extern "C" int gogp_test_dgemm_nt(int device, int64_t M, int64_t N, int64_t K, double alpha,
                                  const double *A, const double *B, double beta, double *C) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return GOGP_EARG;
  if (M % TILE || N % TILE || K % GEMM_BK) return GOGP_EARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  hipError_t e = hipMalloc(&dA, (size_t)M * K * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dB, (size_t)N * K * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dC, (size_t)M * N * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(dA, A, (size_t)M * K * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dB, B, (size_t)N * K * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dC, C, (size_t)M * N * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    launch_dgemm_nt(0, GEMM_RECT, (int)(M / TILE), (int)(N / TILE), K, alpha, dA, K, dB, K, beta,
                    dC, N, nullptr);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(C, dC, (size_t)M * N * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  return e == hipSuccess ? GOGP_OK : GOGP_EHIP;
}

// Diagnostic: factor+invert one 256x256 SPD block (host buffers) with the stamped
// build of the diagonal kernel; returns the factor, the inverse and 24 s_memtime stamps.
extern "C" int gogp_test_diag256(int device, const double *A, double *Lout, double *Dinv,
                                 unsigned long long *stamps, double *elapsed_us) {
  if (!A || !Lout || !Dinv || !stamps) return GOGP_EARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  double *dA = nullptr, *dL = nullptr, *dD = nullptr;
  long long *dinfo = nullptr;
  unsigned long long *dst = nullptr;
  const size_t nb = 256 * 256 * sizeof(double);
  hipError_t e = hipMalloc(&dA, nb);
  if (e == hipSuccess) e = hipMalloc(&dL, nb);
  if (e == hipSuccess) e = hipMalloc(&dD, nb);
  if (e == hipSuccess) e = hipMalloc(&dinfo, 8);
  if (e == hipSuccess) e = hipMalloc(&dst, 32 * 8);
  if (e == hipSuccess) e = hipMemcpy(dA, A, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(dinfo, 0, 8);
  if (e == hipSuccess) e = hipMemset(dst, 0, 32 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  if (e == hipSuccess) {
    gogp_th::launch_diag256_stamped(0, dA, dL, dD, dinfo, dst);  // warm-up
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    gogp::launch_diag256(0, dA, 256, dL, 256, dD, 0, 256, dinfo);  // product build, timed
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    gogp_th::launch_diag256_stamped(0, dA, dL, dD, dinfo, dst);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(Lout, dL, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(Dinv, dD, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(stamps, dst, 32 * 8, hipMemcpyDeviceToHost);
  if (elapsed_us) *elapsed_us = ms * 1e3;
  (void)hipFree(dA); (void)hipFree(dL); (void)hipFree(dD); (void)hipFree(dinfo); (void)hipFree(dst);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return e == hipSuccess ? GOGP_OK : GOGP_EHIP;
}

// ---- instruction costs of the pivot chain (pivot16.h): cycles (s_memtime) per instruction of one wave alone on its SIMD,
// for chains of dependent and runs of independent fp64 operations.  out[0..7]: dependent v_fma_f64, independent v_fma_f64,
// dependent v_mov_b64_dpp row_newbcast, independent v_mov_b64_dpp, dependent v_rsq_f64, independent v_rsq_f64,
// dependent v_mul_f64, dpp -> fma -> dpp -> fma dependent pairs (per pair).
__global__ __launch_bounds__(64) void valu_cost_kernel(double *out, double seed) {
  double x = seed + 1e-9 * threadIdx.x, y = 1.0000001, z = 0.9999999;
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  unsigned long long t[9];
#define GOGP_T(k) t[k] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
  GOGP_T(0);
  asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a0) : "v"(y), "v"(z));
  GOGP_T(1);
  asm volatile(REP8("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                    "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t")
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y), "v"(z));
  GOGP_T(2);
  asm volatile(REP64("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t") : "+v"(a0));
  GOGP_T(3);
  asm volatile(REP8("v_mov_b64_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_mov_b64_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_mov_b64_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_mov_b64_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_mov_b64_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_mov_b64_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                    "v_mov_b64_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t")
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
  GOGP_T(4);
  asm volatile(REP64("v_rsq_f64 %0, %0\n\t") : "+v"(a0));
  GOGP_T(5);
  asm volatile(REP8("v_rsq_f64 %0, %8\n\tv_rsq_f64 %1, %8\n\tv_rsq_f64 %2, %8\n\tv_rsq_f64 %3, %8\n\t"
                    "v_rsq_f64 %4, %8\n\tv_rsq_f64 %5, %8\n\tv_rsq_f64 %6, %8\n\tv_rsq_f64 %7, %8\n\t")
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(y));
  GOGP_T(6);
  asm volatile(REP64("v_mul_f64 %0, %0, %1\n\t") : "+v"(a1) : "v"(y));
  GOGP_T(7);
  asm volatile(REP64("s_nop 1\n\tv_mov_b64_dpp %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_fma_f64 %0, %1, %2, %0\n\t")
               : "+v"(a2), "+v"(a3) : "v"(y));
  GOGP_T(8);
#undef GOGP_T
#undef REP8
#undef REP64
  if (threadIdx.x == 0)
    for (int k = 0; k < 8; ++k) out[k] = (double)(t[k + 1] - t[k]) / 64.0;
  out[8 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
extern "C" int gogp_test_valu_cost(int device, double *out8) {
  if (!out8) return GOGP_EARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  double *d = nullptr;
  if (hipMalloc(&d, (8 + 64) * sizeof(double)) != hipSuccess) return GOGP_ENOMEM;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(valu_cost_kernel, dim3(1), dim3(64), 0, 0, d, 1.5);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out8, d, 8 * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e == hipSuccess ? GOGP_OK : GOGP_EHIP;
}

// Diagnostic: one 128-column chain step (panel128.hip) on a (128 + rows_below) x 128 panel given on the host (row-major,
// ld = 128; the diagonal block's lower triangle is used): the factor of the diagonal block and the solved rows, 72
// s_memtime stamps of the diagnostic build, and the HIP-event time of `reps` launches of the product build.
extern "C" int gogp_test_panel128(int device, const double *A, double *Lout, int64_t rows_below, int reps,
                                  unsigned long long *stamps, double *elapsed_us) {
  if (!A || !Lout || !stamps || rows_below < 0 || rows_below % 64 || reps <= 0) return GOGP_EARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  double *dA = nullptr, *dL = nullptr;
  long long *dinfo = nullptr;
  unsigned long long *dst = nullptr;
  // the product launcher also zeroes the block right of the diagonal block: it gets a 256-wide matrix
  const size_t rows = 128 + (size_t)rows_below, nb = rows * 128 * sizeof(double), nb2 = rows * 256 * sizeof(double);
  // each buffer: the ld = 128 matrix (the stamped build's), then the ld = 256 one (the product launcher's)
  hipError_t e = hipMalloc(&dA, nb + nb2);
  if (e == hipSuccess) e = hipMalloc(&dL, nb + nb2);
  if (e == hipSuccess) e = hipMalloc(&dinfo, 8);
  if (e == hipSuccess) e = hipMalloc(&dst, 72 * 8);
  if (e == hipSuccess) e = hipMemcpy(dA, A, nb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy2D(dA + rows * 128, 256 * sizeof(double), A, 128 * sizeof(double), 128 * sizeof(double), rows,
                                       hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(dinfo, 0, 8);
  if (e == hipSuccess) e = hipMemset(dst, 0, 72 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  if (e == hipSuccess) {
    double *A2 = dA + rows * 128, *L2 = dL + rows * 128;  // the ld = 256 copy (the second halves of both buffers)
    gogp::launch_panel128(0, A2, 256, L2, 256, 0, rows_below, 0, (int64_t)rows, dinfo);  // warm-up
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) gogp::launch_panel128(0, A2, 256, L2, 256, 0, rows_below, 0, (int64_t)rows, dinfo);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    gogp_th::launch_panel128_stamped(0, dA, dL, rows_below, dinfo, dst);
    (void)hipDeviceSynchronize();
    gogp_th::launch_panel128_stamped(0, dA, dL, rows_below, dinfo, dst);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(Lout, dL, nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(stamps, dst, 72 * 8, hipMemcpyDeviceToHost);
  if (elapsed_us) *elapsed_us = ms * 1e3 / reps;
  (void)hipFree(dA); (void)hipFree(dL); (void)hipFree(dinfo); (void)hipFree(dst);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return e == hipSuccess ? GOGP_OK : GOGP_EHIP;
}

// The product build of the chain step with the number of 64-row slabs per workgroup forced (panel128.hip: the result must
// not depend on it), on a (128 + rows_below) x 128 panel given on the host; HIP-event time per launch over `reps` launches.
extern "C" int gogp_test_panel128_slabs(int device, const double *A, double *Lout, int64_t rows_below, int slabs, int reps,
                                        double *elapsed_us) {
  if (!A || !Lout || rows_below < 0 || rows_below % 64 || slabs < 0 || slabs > 8 || reps <= 0) return GOGP_EARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  const size_t rows = 128 + (size_t)rows_below, nb2 = rows * 256 * sizeof(double);
  double *dA = nullptr, *dL = nullptr;
  long long *dinfo = nullptr;
  hipError_t e = hipMalloc(&dA, nb2);
  if (e == hipSuccess) e = hipMalloc(&dL, nb2);
  if (e == hipSuccess) e = hipMalloc(&dinfo, 8);
  if (e == hipSuccess) e = hipMemset(dL, 0, nb2);
  if (e == hipSuccess) e = hipMemset(dinfo, 0, 8);
  if (e == hipSuccess) e = hipMemcpy2D(dA, 256 * sizeof(double), A, 128 * sizeof(double), 128 * sizeof(double), rows, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  if (e == hipSuccess) {
    gogp::launch_panel128_slabs(0, dA, 256, dL, 256, 0, rows_below, 0, (int64_t)rows, dinfo, slabs);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) gogp::launch_panel128_slabs(0, dA, 256, dL, 256, 0, rows_below, 0, (int64_t)rows, dinfo, slabs);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy2D(Lout, 128 * sizeof(double), dL, 256 * sizeof(double), 128 * sizeof(double), rows, hipMemcpyDeviceToHost);
  if (elapsed_us) *elapsed_us = ms * 1e3 / reps;
  (void)hipFree(dA); (void)hipFree(dL); (void)hipFree(dinfo);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return e == hipSuccess ? GOGP_OK : GOGP_EHIP;
}

// Benchmark hook for the tile kernel: times `reps` launches of one GEMM shape on
// device-resident pseudo-random operands (lda = ldb = K, ldc = nt*128).
extern "C" int gogp_bench_gemm(int device, int mode, int mt, int nt, int64_t K, int reps,
                               double *ms_per_launch, double *tflops) {
  if (mt <= 0 || nt <= 0 || K <= 0 || K % GEMM_BK || reps <= 0 || mode < 0 || mode > 2)
    return GOGP_EARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GOGP_EHIP;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GOGP_EHIP;
  const int64_t M = (int64_t)mt * TILE, N = (int64_t)nt * TILE;
  const int64_t Kld = (mode == GEMM_LAUUM) ? M : K;  // LAUUM: K range = matrix size
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  hipError_t e = hipMalloc(&dA, (size_t)M * Kld * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dB, (size_t)N * Kld * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dC, (size_t)M * N * sizeof(double));
  if (e != hipSuccess) return GOGP_ENOMEM;
  launch_fill(0, dA, M * Kld, 0.5);
  launch_fill(0, dB, N * Kld, 0.25);
  launch_fill(0, dC, M * N, 1.0);
  GemmProfile pf;
  pf.on = true;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  if (getenv("GOGP_BENCH_GEMM_F32")) {
    // the fp32 tile kernel on the same buffers read as floats (zero-filled: the values do not matter)
    (void)hipMemsetAsync(dA, 0, (size_t)M * Kld * sizeof(double), 0);
    (void)hipMemsetAsync(dB, 0, (size_t)N * Kld * sizeof(double), 0);
    (void)hipMemsetAsync(dC, 0, (size_t)M * N * sizeof(double), 0);
    GemmProfile pf32;
    pf32.on = true;
    hipEvent_t f0, f1;
    (void)hipEventCreate(&f0);
    (void)hipEventCreate(&f1);
    const float *fA = reinterpret_cast<const float *>(dA), *fB = reinterpret_cast<const float *>(dB);
    float *fC = reinterpret_cast<float *>(dC);
    const double beta32 = (mode == GEMM_LAUUM) ? 0.0 : 1.0;
    for (int w = 0; w < 2; ++w)
      launch_gemm_nt(0, (GemmMode)mode, mt, nt, Kld, -1e-3, fA, Kld, fB, Kld, beta32, fC, N, nullptr, nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(f0, 0);
    for (int r = 0; r < reps; ++r)
      launch_gemm_nt(0, (GemmMode)mode, mt, nt, Kld, -1e-3, fA, Kld, fB, Kld, beta32, fC, N, &pf32, nullptr);
    (void)hipEventRecord(f1, 0);
    hipError_t e32 = hipEventSynchronize(f1);
    float ms32 = 0.f;
    (void)hipEventElapsedTime(&ms32, f0, f1);
    if (ms_per_launch) *ms_per_launch = ms32 / reps;
    if (tflops) *tflops = pf32.flops / (ms32 * 1e-3) / 1e12;
    for (auto ev_ : pf32.pool) (void)hipEventDestroy(ev_);
    (void)hipEventDestroy(f0);
    (void)hipEventDestroy(f1);
    (void)hipFree(dA);
    (void)hipFree(dB);
    (void)hipFree(dC);
    return e32 == hipSuccess ? GOGP_OK : GOGP_EHIP;
  }
  // GOGP_BENCH_GEMM_BETA0=1: beta = 0, i.e. no C-tile read (what the C preload costs a tile)
  const double beta = (mode == GEMM_LAUUM || getenv("GOGP_BENCH_GEMM_BETA0")) ? 0.0 : 1.0;
  // GOGP_BENCH_GEMM_LD0=1: every operand row aliases row 0 (lda = ldb = 0): all operand loads hit in the
  // caches -- the kernel's rate with memory latency taken out (diagnostic)
  const int64_t ld = getenv("GOGP_BENCH_GEMM_LD0") ? 0 : Kld;
  for (int w = 0; w < 2; ++w)
    launch_dgemm_nt(0, (GemmMode)mode, mt, nt, Kld, -1e-3, dA, ld, dB, ld, beta, dC, N, nullptr);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r)
    launch_dgemm_nt(0, (GemmMode)mode, mt, nt, Kld, -1e-3, dA, ld, dB, ld, beta, dC, N, &pf);
  (void)hipEventRecord(e1, 0);
  e = hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_launch) *ms_per_launch = ms / reps;
  if (tflops) *tflops = pf.flops / (ms * 1e-3) / 1e12;
  for (auto ev_ : pf.pool) (void)hipEventDestroy(ev_);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  return e == hipSuccess ? GOGP_OK : GOGP_EHIP;
}
