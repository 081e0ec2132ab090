// grad_mfma.hip -- the fused gradient reduction for ONE radial term with per-dimension (ARD) length
// scales, with both O(N^2 D) sums on the matrix cores.
//
// Reference: gp.GP.Gradient (gp/gp.go:418-499), grad_p = 1/2 tr((alpha alpha^T - K^-1) dK_p) per parameter,
// dK_p built per pair by the AD tape (gp/gp.go:113-117); grad.hip explains the single fused pass
//     grad_p = 1/2 sum_ij W_ij theta_p dk_ij/dtheta_p,  W = alpha alpha^T - K^-1.
// For k = c f(r^2), r^2 = sum_d ((x_id - x_jd) / l_d)^2, the length-scale components are
//     sum_ij g_ij u_ijd^2,   u_ijd = xs_id - xs_jd,  xs = x / l,  g_ij = -2 c W_ij f'(r_ij^2),
// i.e. per pair D subtract-multiply-FMA triples for r^2 and D more for the components: at D = 32 that
// arithmetic (not the 8 N^2 bytes of K^-1) is the whole cost of grad_reduce_kernel (config 5: 64 ms).
// Both sums are GEMM-shaped:
//     r_ij^2       = |xs_i|^2 + |xs_j|^2 - 2 (Xs Xs^T)_ij
//     sum_j g_ij u_ijd^2 = xs_id^2 R_i + (G Xs2)_id - 2 xs_id (G Xs)_id  ... summed over i, with column sums:
//     sum_ij g_ij u_ijd^2 = sum_i xs_id^2 R_i + sum_j xs_jd^2 C_j - 2 sum_i xs_id (G Xs)_id
// (R_i, C_j: row / column sums of the tile of G).  Per 64x64 tile of K^-1 the kernel forms S = Xs_r Xs_c^T
// (64 x 64 x D) and P = G Xs_c (64 x D x 64) with v_mfma_f64_16x16x4_f64 from LDS; what is left per pair on
// the vector ALU is the kernel function itself (one exp) and a handful of multiplies.  All sums are fp64 and
// fixed-order (bitwise reproducible).  The x^2 + y^2 - 2xy forms cancel relative to |xs|^2, so every tile works on
// coordinates centred on its first column point (the kernel is translation invariant): what is squared is then of
// the size of the pair distances, whatever offset the caller's inputs carry (test_ard_gradient_offset_inputs).
#include "kern_eval.h"

namespace gogp {

typedef double f64x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ f64x4 mfma4(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
}  // namespace

// DP: D rounded up to a multiple of 16 (16, 32, 48, 64); LDS rows have DP + 2 doubles (the MFMA fragment
// reads of 16 rows x 2 k then hit 64 distinct banks).
template <int DP, bool LOCAL, class KT>
__global__ __launch_bounds__(256, 2) void grad_ard_mfma_kernel(
    const DevParams *__restrict__ Pp, const double *__restrict__ X, const double *__restrict__ alpha,
    const KT *__restrict__ Kinv, long ld, long n, int nt, int ntiles, double *__restrict__ partials, int ntc,
    BlockMap map, long bstride) {
  constexpr int XS = DP + 2;   // row stride of the coordinate blocks
  constexpr int GS = 66;       // row stride of the G tile
  constexpr int NTD = DP / 16; // 16-column tiles of P
  extern __shared__ double sm[];
  const DevParams &P = *cand(Pp, bstride);  // candidate batching (common.h: Batch); X is shared
  alpha = cand(alpha, bstride);
  Kinv = cand(Kinv, bstride);
  partials = cand(partials, bstride);
  const int D = P.ndim;
  double *Xr = sm;                  // [64][XS] scaled rows
  double *Xc = Xr + 64 * XS;        // [64][XS] scaled columns
  double *Gt = Xc + 64 * XS;        // [64][GS]
  double *nr = Gt + 64 * GS;        // [64] |xs_i|^2
  double *nc = nr + 64;             // [64]
  double *ai = nc + 64;             // [64]
  double *aj = ai + 64;             // [64]
  double *Rg = aj + 64;             // [64] row sums of G
  double *Cw = Rg + 64;             // [4][64] column sums of G per wave
  double *red = Cw + 256;           // [4][NACC] final reduction
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;

  const int kind = P.kind[0];
  const double cc = P.c[0];
  double acc0 = 0.0, trace = 0.0;   // scale component, trace(W)
  double ardp[NTD];                 // -2 sum_i xs_id P_id for d = 16 td + (lane & 15), this lane's rows
#pragma unroll
  for (int td = 0; td < NTD; ++td) ardp[td] = 0.0;
  double ards = 0.0;                // sum_i xs_id^2 R_i + sum_j xs_jd^2 C_j for d = tid & 63, rows 16 (tid >> 6) ..

  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int ti, tj;
    long r0, c0, lr0, lc0;
    if (LOCAL) {
      ti = t / ntc;
      tj = t - ti * ntc;
      lr0 = (long)ti * 64;
      lc0 = (long)tj * 64;
      r0 = map.grow(lr0);
      c0 = map.gcol(lc0);
      if (c0 > r0) continue;  // workgroup-uniform: tile of the global upper triangle
    } else {
      ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
      while (ti * (ti + 1) / 2 > t) --ti;
      while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
      tj = t - ti * (ti + 1) / 2;
      lr0 = r0 = (long)ti * 64;
      lc0 = c0 = (long)tj * 64;
    }
    __syncthreads();  // the previous tile's readers are done
    // ---- scaled coordinates of the tile's rows and columns (zero beyond D and beyond n) ----------------
    // Centred on the tile's first column point: r^2 = |a|^2 + |b|^2 - 2 a.b and the x^2 R + x^2 C - 2 x (G X) sums
    // cancel with a relative error of eps |xs|^2 / u^2, harmless for coordinates of the size of the pair distances
    // but not for uncentred inputs (timestamps: |x / l| ~ 1e5 would leave 1e-6).  The kernel only depends on
    // differences, so the shift is free; x - ref is exact or rounded relative to the DIFFERENCE.
    const long cref = (c0 < n ? c0 : n - 1) * D;
    for (int idx = tid; idx < 64 * DP; idx += 256) {
      const int r = idx / DP, d = idx - r * DP;
      const double il = d < D ? P.inv_len[0][d] : 0.0;
      const double ref = d < D ? X[cref + d] : 0.0;
      Xr[r * XS + d] = (d < D && r0 + r < n) ? (X[(r0 + r) * D + d] - ref) * il : 0.0;
      Xc[r * XS + d] = (d < D && c0 + r < n) ? (X[(c0 + r) * D + d] - ref) * il : 0.0;
    }
    if (tid < 64) ai[tid] = (r0 + tid < n) ? alpha[r0 + tid] : 0.0;
    else if (tid < 128) aj[tid - 64] = (c0 + tid - 64 < n) ? alpha[c0 + tid - 64] : 0.0;
    __syncthreads();
    if (tid < 128) {  // squared norms, one row / column per thread, fixed order
      const double *src = (tid < 64 ? Xr : Xc) + (tid & 63) * XS;
      double s = 0.0;
      for (int d = 0; d < DP; ++d) s += src[d] * src[d];
      (tid < 64 ? nr : nc)[tid & 63] = s;
    }
    // this lane's 16 elements of K^-1, requested before the products so that their latency hides behind
    // them (padded matrix: every address of the tile is valid; inside the per-pair branch below the loads
    // would go out one at a time, each waiting for the previous pair's arithmetic)
    // (not at DP = 64: the 32 extra registers would take the instance past 256, and its 109 KB of LDS allow
    // one workgroup per CU anyway)
    constexpr bool PREFETCH = DP < 64 || sizeof(KT) == 4;
    KT kv[4][4];
    if (PREFETCH) {
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int v = 0; v < 4; ++v) kv[tb][v] = Kinv[(lr0 + 16 * w + fk + 4 * v) * ld + lc0 + 16 * tb + fr];
    }
    // ---- S = Xs_r Xs_c^T: wave w owns rows 16 w .. 16 w + 15, the four 16-column blocks -----------------
    f64x4 sacc[4];
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) sacc[tb] = (f64x4){0.0, 0.0, 0.0, 0.0};
    {
      const double *ap = Xr + (16 * w + fr) * XS + fk;
      const double *bp = Xc + fr * XS + fk;
#pragma unroll
      for (int ks = 0; ks < DP / 4; ++ks) {
        const double a = ap[4 * ks];
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) sacc[tb] = mfma4(a, bp[tb * 16 * XS + 4 * ks], sacc[tb]);
      }
    }
    __syncthreads();  // norms written
    // ---- per pair: weight, kernel function, g; C layout: row = 16 w + fk + 4 v, column = 16 tb + fr ----------
    double rs[4] = {0.0, 0.0, 0.0, 0.0};  // row sums of g over this lane's columns, per v
    double cs[4] = {0.0, 0.0, 0.0, 0.0};  // column sums of g over this lane's rows, per tb
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      const int j = 16 * tb + fr;
      const long gj = c0 + j;
      const double ncj = nc[j], ajv = aj[j];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = 16 * w + fk + 4 * v;
        const long gi = r0 + i;
        double g = 0.0;
        if (gi < n && gj <= gi) {
          double r2 = nr[i] + ncj - 2.0 * sacc[tb][v];
          r2 = r2 > 0.0 ? r2 : 0.0;
          const double wv = ai[i] * ajv - (double)(PREFETCH ? kv[tb][v] : Kinv[(lr0 + i) * ld + lc0 + j]);
          const double wgt = (gj < gi) ? 2.0 * wv : wv;
          double f, dfdr2;
          radial_eval(kind, r2, f, dfdr2);
          acc0 += wgt * cc * f;
          g = wgt * cc * dfdr2 * (-2.0);
          if (gi == gj) trace += wv;
        }
        Gt[i * GS + j] = g;
        rs[v] += g;
        cs[tb] += g;
      }
    }
    // row sums: over the 16 lanes that share fk (xor 1, 2, 4, 8); column sums: over the 4 lanes that share fr
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double s = rs[v];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
      if (fr == 0) Rg[16 * w + fk + 4 * v] = s;
    }
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) {
      double s = cs[tb];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (fk == 0) Cw[w * 64 + 16 * tb + fr] = s;
    }
    __syncthreads();  // G, R, C complete
    // ---- P = G Xs_c (64 x DP): wave w owns rows 16 w .., all DP / 16 column blocks ------------------------
    f64x4 pacc[NTD];
#pragma unroll
    for (int td = 0; td < NTD; ++td) pacc[td] = (f64x4){0.0, 0.0, 0.0, 0.0};
    {
      const double *ap = Gt + (16 * w + fr) * GS + fk;
      const double *bp = Xc + fk * XS + fr;  // B[k = column j][n = dimension] = Xs_c[j][d]
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const double a = ap[4 * ks];
#pragma unroll
        for (int td = 0; td < NTD; ++td) pacc[td] = mfma4(a, bp[4 * ks * XS + 16 * td], pacc[td]);
      }
    }
#pragma unroll
    for (int td = 0; td < NTD; ++td) {
      double s = 0.0;
#pragma unroll
      for (int v = 0; v < 4; ++v) s += Xr[(16 * w + fk + 4 * v) * XS + 16 * td + fr] * pacc[td][v];
      ardp[td] -= 2.0 * s;
    }
    // ---- sum_i xs_id^2 R_i + sum_j xs_jd^2 C_j: thread (d = tid & 63, rows 16 (tid >> 6) .. + 15) -------------
    {
      const int d = tid & 63, q0 = 16 * (tid >> 6);
      if (d < DP) {
        double s = 0.0;
        for (int i = q0; i < q0 + 16; ++i) {
          const double xr = Xr[i * XS + d], xc = Xc[i * XS + d];
          const double cg = (Cw[i] + Cw[64 + i]) + (Cw[128 + i] + Cw[192 + i]);
          s += xr * xr * Rg[i] + xc * xc * cg;
        }
        ards += s;
      }
    }
  }

  // ---- workgroup reduction, fixed order ----------------------------------------------------------------------
  __syncthreads();
  double *slots = Xr;  // [4 waves][NACC] scratch (the coordinate blocks are dead)
  for (int idx = tid; idx < 4 * NACC; idx += 256) slots[idx] = 0.0;
  __syncthreads();
  {
    double v0 = acc0, v1 = trace;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      v0 += __shfl_xor(v0, o);
      v1 += __shfl_xor(v1, o);
    }
    if (lane == 0) {
      slots[w * NACC + 0] = v0;
      slots[w * NACC + ACC_TRACE] = v1;
    }
  }
  // ardp[td]: lanes that share fr hold the same dimension d = 16 td + fr
#pragma unroll
  for (int td = 0; td < NTD; ++td) {
    double s = ardp[td];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (fk == 0) slots[w * NACC + ACC_ARD0 + 16 * td + fr] = s;
  }
  __syncthreads();
  // ards: thread tid holds dimension tid & 63 for row quarter tid >> 6 = its wave: add into the wave's slot
  if ((tid & 63) < DP) slots[w * NACC + ACC_ARD0 + (tid & 63)] += ards;
  __syncthreads();
  if (tid < NACC) {
    const double v = (slots[tid] + slots[NACC + tid]) + (slots[2 * NACC + tid] + slots[3 * NACC + tid]);
    partials[(long)blockIdx.x * NACC + tid] = v;
  }
  (void)red;
}

// LDS bytes of the kernel for DP
static size_t ard_mfma_lds(int DP) {
  return (size_t)(2 * 64 * (DP + 2) + 64 * 66 + 6 * 64 + 256 + 4 * NACC) * sizeof(double);
}

template <bool LOCAL, class KT>
static void launch_ard_mfma_t(hipStream_t s, const DevParams *p, int ndim, const double *X, const double *alpha,
                              const KT *Kinv, int64_t ld, int64_t n, int nt, int ntiles, int blocks, unsigned nz,
                              double *partials, int ntc, BlockMap map, long bstride) {
  const int DP = (ndim + 15) / 16 * 16;
  const size_t lds = ard_mfma_lds(DP);
#define GOGP_LAUNCH_AM(DPV)                                                                              \
  do {                                                                                                   \
    /* > 64 KB of dynamic LDS needs the attribute; set per launch (per device, cheap next to the kernel) */ \
    if (lds > 65536)                                                                                      \
      (void)hipFuncSetAttribute((const void *)grad_ard_mfma_kernel<DPV, LOCAL, KT>,                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    GOGP_KLAUNCH((grad_ard_mfma_kernel<DPV, LOCAL, KT>), dim3(blocks, 1, nz), dim3(256), lds, s, p, X, alpha, \
                       Kinv, (long)ld, (long)n, nt, ntiles, partials, ntc, map, bstride);                 \
  } while (0)
  if (DP == 16) GOGP_LAUNCH_AM(16);
  else if (DP == 32) GOGP_LAUNCH_AM(32);
  else if (DP == 48) GOGP_LAUNCH_AM(48);
  else GOGP_LAUNCH_AM(64);
#undef GOGP_LAUNCH_AM
}

// the reduction pass itself (grad.hip adds the final cross-block sum); ntc == 0: the lower triangle of an
// unsharded K^-1 (nt x nt tiles of 64), else the local nt x ntc tiles of a 2-D block-cyclic one
void launch_grad_ard_mfma(hipStream_t s, const DevParams *p, int ndim, const double *X, const double *alpha,
                          const double *Kinv, int64_t ld, int64_t n, int nt, int ntc, int ntiles, int blocks,
                          BlockMap map, double *partials) {
  if (ntc == 0)
    launch_ard_mfma_t<false, double>(s, p, ndim, X, alpha, Kinv, ld, n, nt, ntiles, blocks, (unsigned)tl_batch.k,
                                     partials, 0, map, tl_batch.stride);
  else
    launch_ard_mfma_t<true, double>(s, p, ndim, X, alpha, Kinv, ld, n, nt, ntiles, blocks, 1u, partials, ntc, map, 0L);
}
void launch_grad_ard_mfma(hipStream_t s, const DevParams *p, int ndim, const double *X, const double *alpha,
                          const float *Kinv, int64_t ld, int64_t n, int nt, int ntc, int ntiles, int blocks,
                          BlockMap map, double *partials) {
  if (ntc == 0)
    launch_ard_mfma_t<false, float>(s, p, ndim, X, alpha, Kinv, ld, n, nt, ntiles, blocks, (unsigned)tl_batch.k,
                                    partials, 0, map, tl_batch.stride);
  else
    launch_ard_mfma_t<true, float>(s, p, ndim, X, alpha, Kinv, ld, n, nt, ntiles, blocks, 1u, partials, ntc, map, 0L);
}

}  // namespace gogp
