// diagsyrk.hip -- fp32 path (BASELINE config 5): the trailing updates of the 256 x 256 DIAGONAL blocks accumulated in fp64.
//
// Reference counterpart: the Dsyrk trailing updates of gonum's blocked Dpotrf (mat.Cholesky.Factorize, gp/gp.go:228), in
// the reference all fp64.  On the fp32 path the trailing matrix is float and its updates run on v_mfma_f32_32x32x2_f32.
// Measured (tools/fp32_bias_probe.py, round 5): the float factor's DIAGONAL comes out systematically too large -- mean
// signed relative error of L_ii +3e-7 per preceding 256-panel against an rms of the same size (1.8e-6 in the seventh
// block of an N = 1721 case), i.e. a bias, not noise: the float accumulation of a - sum l^2 loses on the side of the sum --
// and tr(K^-1) = sum 1/L_ii^2 + ... inherits twice that bias, which is the whole error of the gradient's noise component
// (DESIGN.md section 6).  The pivots are what the fp64 diagonal-block kernel starts from, so they are kept apart: a strip
// D64 of N / 256 blocks of 256 x 256 doubles starts as the (rounded-once) Gram blocks and receives every panel's
// contribution  D64_p -= L[p, k0:k1] L[p, k0:k1]^T  summed in fp64 from the float panel rows (v_mfma_f64_16x16x4_f64 on
// operands widened while they are staged); the diagonal-block kernel factors D64_p instead of the float block.  1/64 of
// the update's tiles at N = 16384, in the half-rate arithmetic: 0.4 % of an evaluation's time at config 5.
#include "common.h"

namespace gogp {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int DS_T = 64;    // tile
constexpr int DS_KC = 16;   // k per staged chunk
constexpr int DS_LD = 18;   // LDS leading dimension: fragment reads of 16 rows x 2 k hit 64 distinct banks

// blockIdx.x: lower 64 x 64 tile (ti >= tj) of the BS x BS block, blockIdx.y: block b -- its BS rows of K float columns start at
// Lrows + b * row_stride; D64 + b * BS^2 (leading dimension BS) -= rows_ti rows_tj^T.  BS = 256: the single-GPU path's
// diagonal blocks (row_stride = 256 ld); BS = 512: the diagonal tiles of a sharded evaluation, whose rows of the panel are
// the rank's own nb x nb tiles of it (dist2d.hip: row_stride = tiles between two diagonal tiles of this rank).
__global__ __launch_bounds__(256, 2) void diag_syrk_kernel(const float *__restrict__ Lrows, long ld, long K,
                                                         double *__restrict__ D64, long row_stride, int BS) {
  __shared__ __attribute__((aligned(16))) double As[2][DS_T * DS_LD], Bs[2][DS_T * DS_LD];
  int t = blockIdx.x, ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
  const int tj = t - ti * (ti + 1) / 2;
  const float *base = Lrows + (long)blockIdx.y * row_stride;
  double *D = D64 + (long)blockIdx.y * BS * BS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  // staging: thread -> row tid / 4, four floats at k = 4 (tid % 4)
  const int sr = tid >> 2, sk = (tid & 3) * 4;
  const float *pa = base + (long)(ti * DS_T + sr) * ld + sk;
  const float *pb = base + (long)(tj * DS_T + sr) * ld + sk;
  f64x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      acc[c][v] = D[(long)(ti * DS_T + 16 * w + fk + 4 * v) * BS + tj * DS_T + 16 * c + fr];
  float4 ra = *reinterpret_cast<const float4 *>(pa), rb = *reinterpret_cast<const float4 *>(pb);
  const int so = sr * DS_LD + sk;
  const long nch = K / DS_KC;
  for (long ch = 0; ch < nch; ++ch) {
    double *as = As[ch & 1], *bs = Bs[ch & 1];
    // the minus sign of the update rides on A
    as[so] = -(double)ra.x; as[so + 1] = -(double)ra.y; as[so + 2] = -(double)ra.z; as[so + 3] = -(double)ra.w;
    bs[so] = (double)rb.x; bs[so + 1] = (double)rb.y; bs[so + 2] = (double)rb.z; bs[so + 3] = (double)rb.w;
    if (ch + 1 < nch) {
      ra = *reinterpret_cast<const float4 *>(pa + (ch + 1) * DS_KC);
      rb = *reinterpret_cast<const float4 *>(pb + (ch + 1) * DS_KC);
    }
    __syncthreads();  // chunk ch staged; chunk ch - 1's buffer (the other one) is free for the next iteration's stores
#pragma unroll
    for (int k4 = 0; k4 < DS_KC / 4; ++k4) {
      const double a = as[(16 * w + fr) * DS_LD + 4 * k4 + fk];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double b = bs[(16 * c + fr) * DS_LD + 4 * k4 + fk];
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      D[(long)(ti * DS_T + 16 * w + fk + 4 * v) * BS + tj * DS_T + 16 * c + fr] = acc[c][v];
}

// D64 block b <- the float block at A + b * 256 * (ld + 1), widened (lower triangle used later; all of it copied)
__global__ __launch_bounds__(256) void widen_diag_blocks_kernel(const float *__restrict__ A, long ld, double *__restrict__ D64) {
  const float *src = A + (long)blockIdx.x * PANEL * (ld + 1);
  double *dst = D64 + (long)blockIdx.x * PANEL * PANEL;
  for (int idx = threadIdx.x; idx < PANEL * PANEL / 4; idx += 256) {
    const int r = idx >> 6, c = (idx & 63) * 4;
    const float4 v = *reinterpret_cast<const float4 *>(src + (long)r * ld + c);
    double *d = dst + r * PANEL + c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

}  // namespace

void launch_diag_syrk_f64(hipStream_t s, const float *Lrows, int64_t ld, int64_t K, double *D64, int nblocks) {
  if (nblocks <= 0 || K <= 0) return;
  GOGP_KLAUNCH(diag_syrk_kernel, dim3(10, (unsigned)nblocks), dim3(256), 0, s, Lrows, (long)ld, (long)K, D64,
               (long)PANEL * (long)ld, (int)PANEL);
}

void launch_diag_syrk_f64_tiles(hipStream_t s, const float *Lrows, int64_t ld, int64_t K, double *D64, int nblocks,
                                int64_t row_stride, int bs) {
  if (nblocks <= 0 || K <= 0) return;
  const int nt = bs / DS_T;
  GOGP_KLAUNCH(diag_syrk_kernel, dim3((unsigned)(nt * (nt + 1) / 2), (unsigned)nblocks), dim3(256), 0, s, Lrows, (long)ld,
               (long)K, D64, (long)row_stride, bs);
}

void launch_widen_diag_blocks(hipStream_t s, const float *A, int64_t ld, double *D64, int nblocks) {
  if (nblocks <= 0) return;
  GOGP_KLAUNCH(widen_diag_blocks_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, A, (long)ld, D64);
}

}  // namespace gogp
