// sgemm.hip -- the tile kernel of the fp32 path (BASELINE config 5): NT GEMM / SYRK / LAUUM on
// v_mfma_f32_32x32x2_f32 (gfx950: exact f32 in / f32 accumulate, 64 FLOP/clk/SIMD = 157 TFLOP/s).
//
//   C(128x128 tile) = beta*C + alpha * A(128xK) * B(128xK)^T      (row-major, f32)
//
// Same roles and modes as dgemm.hip (trailing update, panel solve through the block inverse,
// triangular inverse, K^-1 = Y Y^T); reference counterpart: gonum's Dpotrf / Dpotri / Dgemm
// behind mat.Cholesky (call sites gp/gp.go:228,338,454,480), here in single precision with the
// diagonal blocks factored in fp64 (diag256.hip) -- see DESIGN.md "fp32 path".
//
// Structure: 128x128 tile, 256 threads = 2x2 waves of 64x64 outputs = 2x2 MFMA 32x32 tiles
// (64 accumulator VGPRs).  K is walked in steps of 32 floats -- one 128-B line per row, the same
// LDS image as the fp64 kernel: operands go global -> LDS directly (global_load_lds_dwordx4),
// double-buffered, 16-B chunks XOR-swizzled with (row >> 1) & 7 (applied on the global side).
// Fragment reads are ds_read_b128: a lane takes FOUR consecutive k of its row at once and feeds
// them to four consecutive MFMAs; lane half h = lane >> 5 reads chunk 2c + h, so MFMA j of chunk
// pair c multiplies k = 8c + 4h + j on BOTH operands (the k order inside a K-step is
// permuted identically for A and B; a sum over k does not care).  16 distinct rows mod 16 per
// ds_read_b128 lane group x the swizzle = 64 distinct banks: conflict-free.
#include <algorithm>

#include <hip/hip_ext.h>

#include "common.h"

namespace gogp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SGEMM_BK = 32;  // floats per K-step = one 128-B line per row

struct SGemmArgs {
  const float *A;
  const float *B;
  float *C;
  long lda, ldb, ldc;
  int mt, nt;
  int nkt;  // K / 32
  float alpha, beta;
  int kend;
  int trap;
  int rule, tpb_shift, rblk0, cblk0, pr, Pr, pc, Pc, beta0;  // GemmGrid, see dgemm.hip
  int new_row0;  // GEMM_LOWER: tile rows >= new_row0 overwrite C (common.h: GemmGrid); INT_MAX: none
  int ktri;      // GEMM_RECT: B lower triangular, tile column tj sums k < (tj + 1) * BT only
  int prio;      // chain launch: s_setprio 3 (common.h: GemmGrid)
  int krag0;     // RECT / LOWER: tile rows ti >= krag0 start at k = (ti - krag0) * BT (common.h: GemmGrid); INT_MAX: none
};

__device__ __forceinline__ void sload16_to_lds(const float *gsrc, float *lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gsrc, lds_wave_base, 16, 0, 0);
#endif
}
__device__ __forceinline__ void sraise_wave_priority() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setprio(3);
#endif
}
__device__ __forceinline__ void swait_vmcnt0() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#endif
}

// BT = workgroup tile (128 or 64).  NW = 4: 2x2 waves, each (BT/2)x(BT/2) outputs = MT x MT MFMA 32x32
// tiles, MT = BT/64.  NW = 8 (BT = 128): 2x4 waves, each 64x32 outputs = 2 x 1 MFMA tiles: half the
// accumulators per wave, four waves per SIMD with two workgroups per CU (the shape of the fp64 kernel's
// large launches, dgemm.hip).
template <int MODE, int BT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void sgemm_nt_kernel(SGemmArgs g) {
  constexpr int MT = BT / 64;                          // MFMA tiles per wave, rows
  constexpr int NTW = (NW == 8) ? 1 : BT / 64;         // MFMA tiles per wave, columns
  constexpr int WT = BT / 2;                           // rows per wave
  constexpr int WTN = (NW == 8) ? BT / 4 : BT / 2;     // columns per wave
  constexpr int NQ = BT * 8 / (NW * 64);               // staging loads per thread per operand
  constexpr int SROWS = NW * 8;                        // rows one staging pass covers
  __shared__ __attribute__((aligned(16))) float lds[2][2][BT * SGEMM_BK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (g.prio) sraise_wave_priority();

  int t = blockIdx.x;
  if (MODE != GEMM_LAUUM && !(MODE == GEMM_RECT && g.rule)) {  // XCD-aware remap (dgemm.hip)
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  int ti, tj;
  if (MODE == GEMM_RECT && g.rule) {  // filtered launch: tile rows dealt cyclically to the XCDs (dgemm.hip)
    const int x = t & 7, slot = t >> 3;
    const int rr = slot / g.nt;
    ti = x + 8 * rr;
    tj = slot - rr * g.nt;
    if (ti >= g.mt) return;
  } else if (MODE == GEMM_RECT) {
    ti = t / g.nt;
    tj = t - ti * g.nt;
    if (g.trap && (tj * BT) / PANEL > (ti * BT) / PANEL) return;
  } else {
    ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
  }
  float beta = g.beta;
  if (MODE == GEMM_RECT && g.rule) {
    const int gI = (g.rblk0 + (ti >> g.tpb_shift)) * g.Pr + g.pr;
    const int gJ = (g.cblk0 + (tj >> g.tpb_shift)) * g.Pc + g.pc;
    const int msk = (1 << g.tpb_shift) - 1;
    if (gI < gJ || (gI == gJ && (ti & msk) < (tj & msk))) return;
    if (g.rule == 2) beta = (gI == g.beta0) ? 0.0f : 1.0f;
  }
  if (MODE == GEMM_LOWER && ti >= g.new_row0) beta = 0.0f;
  int kbeg = 0, nkt = g.nkt;
  if (MODE == GEMM_RECT && g.ktri) nkt = min(nkt, (tj + 1) * BT / SGEMM_BK);
  if (MODE != GEMM_LAUUM && ti > g.krag0) {
    kbeg = (ti - g.krag0) * BT;
    nkt -= kbeg / SGEMM_BK;
  }
  if (MODE == GEMM_LAUUM) {
    kbeg = ti * BT;
    nkt = (g.kend - kbeg) / SGEMM_BK;
    if (nkt <= 0) return;
  }
  const float *Ag = g.A + (long)ti * BT * g.lda + kbeg;
  const float *Bg = g.B + (long)tj * BT * g.ldb + kbeg;

  // staging: thread -> (row, 16-B chunk); chunk swizzled on the global side
  const int srow = tid >> 3;
  const int schunk = tid & 7;
  const int gchunk = schunk ^ ((srow >> 1) & 7);  // SROWS*q (multiples of 32) never change the swizzle: srow < SROWS, and (srow + SROWS * q) >> 1 & 7 == srow >> 1 & 7 for SROWS = 32 or 64
  const float *Ap = Ag + (long)srow * g.lda + gchunk * 4;
  const float *Bp = Bg + (long)srow * g.ldb + gchunk * 4;
  const long a_step = (long)SROWS * g.lda, b_step = (long)SROWS * g.ldb;

  // fragments: lane -> row (lane & 31) of its MFMA tile, k-half (lane >> 5)
  const int wr = (NW == 8) ? wid >> 2 : wid >> 1;
  const int wc = (NW == 8) ? wid & 3 : wid & 1;
  const int frow = lane & 31, fh = lane >> 5;
  const int abase = (wr * WT + frow) * SGEMM_BK;
  const int bbase = (wc * WTN + frow) * SGEMM_BK;
  int xc[4];  // LDS float offset of chunk 2c + fh of this lane's row (swizzle depends on frow only:
              // wave / MFMA-tile row offsets are multiples of 32)
#pragma unroll
  for (int c = 0; c < 4; ++c) xc[c] = ((2 * c + fh) ^ ((frow >> 1) & 7)) << 2;

  // C fragment of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  float *Cg = g.C + (long)(ti * BT + wr * WT) * g.ldc + tj * BT + wc * WTN;
  const int coff = (4 * fh) * (int)g.ldc + frow;
  const float alpha = g.alpha;

#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    sload16_to_lds(Ap + q * a_step, &lds[0][0][(wid * 8 + SROWS * q) * SGEMM_BK]);
    sload16_to_lds(Bp + q * b_step, &lds[0][1][(wid * 8 + SROWS * q) * SGEMM_BK]);
  }
  f32x16 acc[MT][NTW];
  if (beta != 0.0f) {
    const float sc = beta / alpha;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int v = 0; v < 16; ++v)
          acc[m][n][v] = sc * (Cg + (long)(m * 32 + (v & 3) + 8 * (v >> 2)) * g.ldc)[coff + n * 32];
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[m][n][v] = 0.0f;
  }
  swait_vmcnt0();
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = (kt + 1 < nkt);
    if (more) {
      const float *ap = Ap + (long)(kt + 1) * SGEMM_BK;
      const float *bp = Bp + (long)(kt + 1) * SGEMM_BK;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        sload16_to_lds(ap + q * a_step, &lds[cur ^ 1][0][(wid * 8 + SROWS * q) * SGEMM_BK]);
        sload16_to_lds(bp + q * b_step, &lds[cur ^ 1][1][(wid * 8 + SROWS * q) * SGEMM_BK]);
      }
    }
    const float *la = lds[cur][0];
    const float *lb = lds[cur][1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 a[MT], b[NTW];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const f32x4 *>(la + abase + m * 32 * SGEMM_BK + xc[c]);
#pragma unroll
      for (int n = 0; n < NTW; ++n) b[n] = *reinterpret_cast<const f32x4 *>(lb + bbase + n * 32 * SGEMM_BK + xc[c]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][j], b[n][j], acc[m][n], 0, 0, 0);
    }
    if (more) swait_vmcnt0();
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int v = 0; v < 16; ++v)
        (Cg + (long)(m * 32 + (v & 3) + 8 * (v >> 2)) * g.ldc)[coff + n * 32] = alpha * acc[m][n][v];
}

void launch_gemm_nt(hipStream_t s, GemmMode mode, int mt, int nt, int64_t K, double alpha, const float *A,
                    int64_t lda, const float *B, int64_t ldb, double beta, float *C, int64_t ldc,
                    GemmProfile *prof, const GemmGrid *grid) {
  if (mt <= 0 || nt <= 0 || K <= 0) return;
  SGemmArgs g;
  g.A = A;
  g.B = B;
  g.C = C;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.mt = mt;
  g.nt = nt;
  g.nkt = (int)(K / SGEMM_BK);
  g.alpha = (float)alpha;
  g.beta = (float)beta;
  g.kend = (int)K;
  g.rule = 0;
  g.tpb_shift = g.rblk0 = g.cblk0 = g.pr = g.pc = g.beta0 = 0;
  g.Pr = g.Pc = 1;
  g.new_row0 = (mode == GEMM_LOWER && grid && grid->new_row0 >= 0) ? grid->new_row0 : 0x7fffffff;
  g.ktri = (mode == GEMM_RECT && grid && grid->ktri) ? 1 : 0;
  g.prio = grid ? grid->prio : 0;
  g.krag0 = (mode != GEMM_LAUUM && mode != GEMM_TRAP && grid && grid->krag0 >= 0 && !g.ktri) ? grid->krag0 : 0x7fffffff;
  if (grid && grid->rule) {
    g.rule = grid->rule;
    g.tpb_shift = grid->tpb_shift;
    g.rblk0 = grid->rblk0;
    g.cblk0 = grid->cblk0;
    g.pr = grid->pr;
    g.Pr = grid->Pr;
    g.pc = grid->pc;
    g.Pc = grid->Pc;
    g.beta0 = grid->beta0;
  }
  int ntiles;
  double flops;
  g.trap = 0;
  if (mode == GEMM_TRAP) {
    mode = GEMM_RECT;
    g.trap = 1;
    ntiles = mt * nt;
    const int nb = nt / 2;
    flops = 2.0 * TILE * TILE * (double)K * ((double)mt * nt - 4.0 * nb * (nb - 1) / 2.0);
  } else if (mode == GEMM_RECT) {
    ntiles = mt * nt;
    flops = 2.0 * (double)mt * TILE * (double)nt * TILE * (double)K;
    if (g.ktri) {
      flops = 0;
      for (int j = 0; j < nt; ++j)
        flops += 2.0 * (double)mt * TILE * TILE * (double)std::min<int64_t>(K, (int64_t)(j + 1) * TILE);
    }
    if (g.krag0 != 0x7fffffff) {
      flops = 0;
      for (int i = 0; i < mt; ++i)
        flops += 2.0 * (double)nt * TILE * TILE * (double)(K - (int64_t)std::max(0, i - g.krag0) * TILE);
    }
    if (g.rule) {
      const int tpb = 1 << g.tpb_shift;
      long kept = 0;
      for (int bi = 0; bi < mt / tpb; ++bi)
        for (int bj = 0; bj < nt / tpb; ++bj) {
          const int gI = (g.rblk0 + bi) * g.Pr + g.pr, gJ = (g.cblk0 + bj) * g.Pc + g.pc;
          kept += gI > gJ ? (long)tpb * tpb : (gI == gJ ? (long)tpb * (tpb + 1) / 2 : 0);
        }
      flops = 2.0 * (double)kept * TILE * TILE * (double)K;
    }
  } else {
    ntiles = mt * (mt + 1) / 2;
    if (mode == GEMM_LOWER && g.krag0 != 0x7fffffff) {
      flops = 0;
      for (int i = 0; i < mt; ++i)
        flops += 2.0 * (double)(i + 1) * TILE * TILE * (double)(K - (int64_t)std::max(0, i - g.krag0) * TILE);
    } else if (mode == GEMM_LOWER) {
      flops = 2.0 * (double)ntiles * TILE * TILE * (double)K;
    } else {
      flops = 0;
      for (int i = 0; i < mt; ++i)
        flops += 2.0 * (double)(i + 1) * TILE * TILE * (double)(K - (int64_t)i * TILE);
    }
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof && prof->on) {
    if (prof->used + 2 > prof->pool.size()) {
      size_t old = prof->pool.size();
      prof->pool.resize(old + 1024);
      for (size_t i = old; i < prof->pool.size(); ++i) (void)hipEventCreate(&prof->pool[i]);
    }
    e0 = prof->pool[prof->used++];
    e1 = prof->pool[prof->used++];
    prof->flops += flops;
    prof->launches += 1;
    prof->lflops.push_back(flops);
    prof->ltag.push_back((int64_t)mode * 100000000LL + (int64_t)(K / 16) * 100000LL + (int64_t)(ntiles > 99999 ? 99999 : ntiles));
  }
  // With profiling on, the two events ride on the kernel's own dispatch packet
  // (hipExtLaunchKernelGGL: start / stop timestamps of exactly this dispatch) instead of two
  // extra barrier packets in the queue -- the instrumented run keeps the un-instrumented timing.
#define GOGP_LAUNCH(GRID, BLOCK, ...)                                                     \
  do {                                                                                    \
    if (e0)                                                                               \
      hipExtLaunchKernelGGL((__VA_ARGS__), (GRID), (BLOCK), 0, s, e0, e1, 0, g);           \
    else                                                                                  \
      hipLaunchKernelGGL((__VA_ARGS__), (GRID), (BLOCK), 0, s, g);                         \
  } while (0)
  const long wave8_min = 3072;  // as in dgemm.hip; the shape matters less here (N = 32768: 314.1 -> 312.5 ms)
  const bool small = (mode != GEMM_LAUUM) && (ntiles < (grid ? grid->small_below : 384));
  // chain_prio = 1: only the skinny launches (64x64 tiles) raise their priority; 2: every chain launch
  if (g.prio == 1 && !small) g.prio = 0;
  if (small) {  // 64x64 tiles for the skinny GEMMs of the panel chain
    g.mt = mt * 2;
    g.nt = nt * 2;
    g.tpb_shift += 1;
    if (g.new_row0 != 0x7fffffff) g.new_row0 *= 2;
    if (g.krag0 != 0x7fffffff) g.krag0 *= 2;
    const int n64 = (mode == GEMM_RECT) ? (g.rule ? 8 * ((g.mt + 7) / 8) * g.nt : g.mt * g.nt)
                                        : g.mt * (g.mt + 1) / 2;
    if (mode == GEMM_RECT)
      GOGP_LAUNCH(dim3(n64), dim3(256), sgemm_nt_kernel<GEMM_RECT, 64, 4>);
    else
      GOGP_LAUNCH(dim3(n64), dim3(256), sgemm_nt_kernel<GEMM_LOWER, 64, 4>);
  } else if (mode == GEMM_LAUUM || ntiles >= wave8_min) {
    const dim3 gridd(g.rule ? 8 * ((mt + 7) / 8) * nt : ntiles), block8(512);
    if (mode == GEMM_RECT)
      GOGP_LAUNCH(gridd, block8, sgemm_nt_kernel<GEMM_RECT, 128, 8>);
    else if (mode == GEMM_LOWER)
      GOGP_LAUNCH(gridd, block8, sgemm_nt_kernel<GEMM_LOWER, 128, 8>);
    else
      GOGP_LAUNCH(gridd, block8, sgemm_nt_kernel<GEMM_LAUUM, 128, 8>);
  } else {
    const dim3 gridd(g.rule ? 8 * ((mt + 7) / 8) * nt : ntiles), block(256);
    if (mode == GEMM_RECT)
      GOGP_LAUNCH(gridd, block, sgemm_nt_kernel<GEMM_RECT, 128, 4>);
    else
      GOGP_LAUNCH(gridd, block, sgemm_nt_kernel<GEMM_LOWER, 128, 4>);
  }
#undef GOGP_LAUNCH
}

// the fp64 kernel under the same overloaded name (orchestration code is written once for both)
void launch_gemm_nt(hipStream_t s, GemmMode mode, int mt, int nt, int64_t K, double alpha, const double *A,
                    int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc,
                    GemmProfile *prof, const GemmGrid *grid) {
  launch_dgemm_nt(s, mode, mt, nt, K, alpha, A, lda, B, ldb, beta, C, ldc, prof, grid);
}

}  // namespace gogp
