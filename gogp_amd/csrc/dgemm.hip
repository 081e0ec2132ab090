// dgemm.hip -- the dominant kernel: fp64 NT GEMM/SYRK tile kernel on
// v_mfma_f64_16x16x4_f64 (gfx950).
//
//   C(128x128 tile) = beta*C + alpha * A(128xK) * B(128xK)^T      (row-major)
//
// One kernel shape serves every O(N^3) step of the hot path:
//   Cholesky trailing update   A22 -= L21 L21^T        GEMM_LOWER, A == B
//   panel TRSM via the inverse L21  = A21 inv(L11)^T   GEMM_RECT
//   triangular inverse (Y=L^-T) R  -= Y_m L_m^T        GEMM_RECT
//   K^-1 = Y Y^T               ragged K range per tile GEMM_LAUUM
//   Produce: V^T solve         same two RECT forms
//
// Reference counterpart: gonum's Dpotrf/Dpotri/Dtrsm/Dgemm behind
// mat.Cholesky.Factorize / SolveTo (call sites gp/gp.go:228,338,454,480).
//
// Structure (CDNA4): a 128x128 tile per workgroup in two wave shapes.  Large launches
// use 512 threads = 8 waves of 64x32 outputs (4x2 MFMA 16x16 accumulators, 64 acc
// VGPRs, <= 128 VGPRs in all): two workgroups per CU put FOUR waves on every SIMD,
// which hides the barrier / prologue / epilogue bubbles of any one of them.  Mid-size
// launches use 256 threads = 4 waves of 64x64 (128 acc VGPRs, two waves per SIMD), and
// the skinny GEMMs of the panel chain a 64x64 tile.  K is walked in steps of 16:
// both operand tiles (128 rows x 16 doubles = one 128-B line per row) are
// staged global -> registers -> LDS, double-buffered, one barrier per step.
// LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 so
// that the MFMA fragment reads (16 rows x 2 k per 32-lane group, ds_read_b64)
// hit 32 distinct 8-B bank pairs: conflict-free.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace gogp {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct GemmArgs {
  const double *A;
  const double *B;
  double *C;
  long lda, ldb, ldc;
  int mt, nt;
  int nkt;  // K / 16
  double alpha, beta;
  // GEMM_LAUUM only: the K range of tile (ti,tj) is [max(ti*BT, kskip), kend); tiles
  // whose range starts below kskip accumulate (beta = 1), the others overwrite
  int kskip, kend;
  // Ownership filter of a sharded evaluation (own_n <= 1: none).  RECT / LOWER: a tile
  // is computed iff its block column belongs to this rank, i.e.
  // ((own_col0 + tj) / own_tps) % own_n == own_r  (tile columns counted in units of
  // this kernel's tile, own_tps tiles per super-panel).  LAUUM: iff ti % own_n == own_r.
  int own_n, own_r, own_tps, own_col0;
  int trap;  // GEMM_TRAP: skip tiles of strictly upper 256-blocks
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
  // doubles; chunk = 16-B chunk index 0..7 within the 128-B row
  return row * GEMM_BK + ((chunk ^ ((row >> 1) & 7)) << 1);
}

// device-only builtins behind helpers: in the host pass of hipcc the unknown builtin
// silently suppresses the kernel's host stub (undefined __device_stub__ at load time)
__device__ __forceinline__ void load16_to_lds(const double *gsrc, double *lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gsrc, lds_wave_base, 16, 0, 0);
#endif
}
__device__ __forceinline__ void wait_vmcnt0() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt / expcnt untouched
#endif
}

// BT = workgroup tile (128 or 64).  NW = 4: 2x2 waves, each (BT/2)x(BT/2) outputs =
// MT x MT MFMA tiles, MT = BT/32.  NW = 8 (BT = 128): 2x4 waves, each 64x32 outputs:
// half the accumulators per wave (<= 128 VGPRs), so four waves fit on a SIMD.
// DIRECT: operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no
// ds_write): a wave's 64 lanes x 16 B land on 1 KB of consecutive LDS = 8 rows of the tile,
// so the XOR swizzle is applied on the GLOBAL side (lane (row, pos) fetches chunk pos ^ swz(row)).
template <int MODE, int BT, int NW, bool DIRECT>
__global__ __launch_bounds__(NW * 64, NW / 2) void dgemm_nt_kernel(GemmArgs g) {
  constexpr int MT = BT / 32;                          // MFMA tiles per wave, rows
  constexpr int NTW = (NW == 8) ? BT / 64 : BT / 32;   // MFMA tiles per wave, columns
  constexpr int WT = BT / 2;                           // rows per wave
  constexpr int WTN = (NW == 8) ? BT / 4 : BT / 2;     // columns per wave
  constexpr int NQ = BT * 8 / (NW * 64);               // staging loads per thread per operand
  constexpr int SROWS = NW * 8;                        // rows one staging pass covers
  __shared__ __attribute__((aligned(16))) double lds[2][2][BT * GEMM_BK];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile bases stay in SGPRs

  // ---- tile assignment ----------------------------------------------------
  int t = blockIdx.x;
  if (MODE != GEMM_LAUUM) {
    // XCD-aware remap (blocks b and b+8 share an XCD/L2): give each XCD a
    // contiguous chunk of the tile list; bijective for any grid size.
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  int ti, tj;
  if (MODE == GEMM_RECT) {
    ti = t / g.nt;
    tj = t - ti * g.nt;
    if (g.trap && (tj * BT) / PANEL > (ti * BT) / PANEL) return;  // whole-workgroup exit
  } else {
    ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
  }
  if (g.own_n > 1) {  // workgroup-uniform early exit for tiles of other ranks
    if (MODE == GEMM_LAUUM) {
      if (ti % g.own_n != g.own_r) return;
    } else if (((g.own_col0 + tj) / g.own_tps) % g.own_n != g.own_r) {
      return;
    }
  }
  int kbeg = 0, nkt = g.nkt;
  double beta = g.beta;
  if (MODE == GEMM_LAUUM) {
    kbeg = ti * BT;
    if (kbeg < g.kskip) {  // second pass of a split LAUUM: [0, kskip) was summed before
      kbeg = g.kskip;
      beta = 1.0;
    }
    nkt = (g.kend - kbeg) / GEMM_BK;
    if (nkt <= 0) return;  // whole-workgroup exit (tile-uniform)
  }

  const double *Ag = g.A + (long)ti * BT * g.lda + kbeg;
  const double *Bg = g.B + (long)tj * BT * g.ldb + kbeg;

  // ---- staging map: thread -> (row, 16-B chunk), NQ rows per operand --------
  const int srow = tid >> 3;  // 0..SROWS-1, +SROWS*q
  const int schunk = tid & 7;
  const int gchunk = DIRECT ? (schunk ^ ((srow >> 1) & 7)) : schunk;  // SROWS*q never changes swz
  const double *Ap = Ag + (long)srow * g.lda + gchunk * 2;
  const double *Bp = Bg + (long)srow * g.ldb + gchunk * 2;
  const long a_step = (long)SROWS * g.lda, b_step = (long)SROWS * g.ldb;
  int soff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) soff[q] = lds_off(srow + SROWS * q, schunk);

  // ---- fragment map --------------------------------------------------------
  const int wr = (NW == 8) ? wid >> 2 : wid >> 1;
  const int wc = (NW == 8) ? wid & 3 : wid & 1;
  const int frow = lane & 15;
  const int fk = lane >> 4;      // k within an MFMA step: 0..3
  const int fchunk = fk >> 1;    // + 2*kk
  const int fhalf = fk & 1;
  // LDS offset of fragment (m, kk) = base + m * (16 rows) + xk[kk]: the swizzle term
  // (row >> 1) & 7 only depends on frow (wave and MFMA-tile row offsets are multiples
  // of 16), so the m / n steps are immediate offsets of the ds_read
  const int abase = (wr * WT + frow) * GEMM_BK;
  const int bbase = (wc * WTN + frow) * GEMM_BK;
  int xk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xk[kk] = (((kk * 2 + fchunk) ^ ((frow >> 1) & 7)) << 1) + fhalf;

  // C fragment of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  double *Cg = g.C + (long)(ti * BT + wr * WT) * g.ldc + tj * BT + wc * WTN;
  // address = wave-uniform row base (SGPRs) + ONE 32-bit per-lane offset + immediate: per-lane
  // 64-bit row pointers kept across the k loop spill, and a spill reload in the epilogue
  // waits (vmcnt counts stores too) for every C store issued before it
  const int coff = (lane >> 4) * (int)g.ldc + (lane & 15);
  const double alpha = g.alpha;

  f64x2 ra[NQ], rb[NQ];
  // prologue loads of k-tile 0 go out first, the C tile right behind them: the
  // accumulators start at (beta/alpha) C, so the epilogue is stores only and the
  // C read latency overlaps the operand pipeline fill
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (DIRECT) {
      load16_to_lds(Ap + q * a_step, &lds[0][0][(wid * 8 + SROWS * q) * GEMM_BK]);
      load16_to_lds(Bp + q * b_step, &lds[0][1][(wid * 8 + SROWS * q) * GEMM_BK]);
    } else {
      ra[q] = *reinterpret_cast<const f64x2 *>(Ap + q * a_step);
      rb[q] = *reinterpret_cast<const f64x2 *>(Bp + q * b_step);
    }
  }
  f64x4 acc[MT][NTW];
  if (beta != 0.0) {
    const double sc = beta / alpha;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          acc[m][n][v] = sc * (Cg + (long)(m * 16 + 4 * v) * g.ldc)[coff + n * 16];
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NTW; ++n) acc[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
  }
  if (!DIRECT) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      *reinterpret_cast<f64x2 *>(&lds[0][0][soff[q]]) = ra[q];
      *reinterpret_cast<f64x2 *>(&lds[0][1][soff[q]]) = rb[q];
    }
  } else {
    wait_vmcnt0();  // the tile is in LDS
  }
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = (kt + 1 < nkt);
    if (more) {
      const double *ap = Ap + (long)(kt + 1) * GEMM_BK;
      const double *bp = Bp + (long)(kt + 1) * GEMM_BK;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (DIRECT) {
          // buffer cur^1 was last read in step kt-1; every wave has passed that step's barrier
          load16_to_lds(ap + q * a_step, &lds[cur ^ 1][0][(wid * 8 + SROWS * q) * GEMM_BK]);
          load16_to_lds(bp + q * b_step, &lds[cur ^ 1][1][(wid * 8 + SROWS * q) * GEMM_BK]);
        } else {
          ra[q] = *reinterpret_cast<const f64x2 *>(ap + q * a_step);
          rb[q] = *reinterpret_cast<const f64x2 *>(bp + q * b_step);
        }
      }
    }
    const double *la = lds[cur][0];
    const double *lb = lds[cur][1];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[MT], b[NTW];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = la[abase + m * 16 * GEMM_BK + xk[kk]];
#pragma unroll
      for (int n = 0; n < NTW; ++n) b[n] = lb[bbase + n * 16 * GEMM_BK + xk[kk]];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    if (more) {
      if (DIRECT) {
        wait_vmcnt0();
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          *reinterpret_cast<f64x2 *>(&lds[cur ^ 1][0][soff[q]]) = ra[q];
          *reinterpret_cast<f64x2 *>(&lds[cur ^ 1][1][soff[q]]) = rb[q];
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: stores only
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        (Cg + (long)(m * 16 + 4 * v) * g.ldc)[coff + n * 16] = alpha * acc[m][n][v];
}

// GOGP_GEMM_W8 = 0 / 1 forces the 4-wave / 8-wave 128-tile kernel (A/B measurements);
// unset: 8 waves for the large launches, where they measured 4-9% faster.
// GOGP_GEMM_DIRECT=0 selects the register-staged operand path (A/B measurements); default:
// global -> LDS direct loads (+3-4 % on the large GEMMs, -2.3 % time per evaluation).
static const bool g_gemm_direct = [] {
  const char *e = getenv("GOGP_GEMM_DIRECT");
  return !e || atoi(e) != 0;
}();

static const int g_gemm_w8 = [] {
  const char *e = getenv("GOGP_GEMM_W8");
  return e ? (atoi(e) != 0 ? 1 : 0) : -1;
}();

void launch_dgemm_nt(hipStream_t s, GemmMode mode, int mt, int nt, int64_t K, double alpha,
                     const double *A, int64_t lda, const double *B, int64_t ldb,
                     double beta, double *C, int64_t ldc, GemmProfile *prof, int64_t kskip,
                     int64_t kend, const GemmOwn *own) {
  if (mt <= 0 || nt <= 0 || K <= 0) return;
  GemmArgs g;
  g.A = A;
  g.B = B;
  g.C = C;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.mt = mt;
  g.nt = nt;
  g.nkt = (int)(K / GEMM_BK);
  g.alpha = alpha;
  g.beta = beta;
  if (kend <= 0 || kend > K) kend = K;
  g.kskip = (int)kskip;
  g.kend = (int)kend;
  g.own_n = own ? own->n : 0;
  g.own_r = own ? own->r : 0;
  g.own_tps = own ? own->tiles_per_sp : 1;
  g.own_col0 = own ? own->col0 : 0;
  int ntiles;
  double flops;
  g.trap = 0;
  if (mode == GEMM_TRAP) {  // rectangular enumeration, upper 256-blocks skipped in the kernel
    mode = GEMM_RECT;
    g.trap = 1;
    ntiles = mt * nt;
    const int nb = nt / 2;  // 256-blocks across; block column b skips b block rows of 2 x 2 tiles
    flops = 2.0 * TILE * TILE * (double)K * ((double)mt * nt - 4.0 * nb * (nb - 1) / 2.0);
  } else if (mode == GEMM_RECT) {
    ntiles = mt * nt;
    flops = 2.0 * (double)mt * TILE * (double)nt * TILE * (double)K;
  } else {
    ntiles = mt * (mt + 1) / 2;
    if (mode == GEMM_LOWER) {
      flops = 2.0 * (double)ntiles * TILE * TILE * (double)K;
    } else {
      flops = 0;
      for (int i = 0; i < mt; ++i) {
        const int64_t kb = std::max<int64_t>((int64_t)i * TILE, kskip);
        if (kend > kb) flops += 2.0 * (double)(i + 1) * TILE * TILE * (double)(kend - kb);
      }
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
    (void)hipEventRecord(e0, s);
  }
  // Small launches (the skinny GEMMs of the panel chain) use 64x64 tiles: 4x the
  // workgroups and a quarter of the per-tile latency.  LAUUM keeps 128 (its K
  // ranges are cut at 128-row granularity).
  static const int small_limit = getenv("GOGP_SMALL_TILES") ? atoi(getenv("GOGP_SMALL_TILES")) : 384;
  const bool small = (mode != GEMM_LAUUM) && (ntiles < small_limit);
  dim3 block(256);
  if (small) {
    g.mt = mt * 2;
    g.nt = nt * 2;
    g.own_tps *= 2;  // ownership counted in 64-wide tile columns
    g.own_col0 *= 2;
    const int n64 = (mode == GEMM_RECT) ? g.mt * g.nt : g.mt * (g.mt + 1) / 2;
    dim3 grid(n64);
    if (mode == GEMM_RECT) {
      if (g_gemm_direct)
        hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_RECT, 64, 4, true>), grid, block, 0, s, g);
      else
        hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_RECT, 64, 4, false>), grid, block, 0, s, g);
    } else {
      if (g_gemm_direct)
        hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LOWER, 64, 4, true>), grid, block, 0, s, g);
      else
        hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LOWER, 64, 4, false>), grid, block, 0, s, g);
    }
  } else if (g_gemm_w8 == 1 || (g_gemm_w8 < 0 && (mode == GEMM_LAUUM || ntiles >= 3072))) {
    dim3 grid(ntiles), block8(512);
    switch (mode) {
      case GEMM_RECT:
        if (g_gemm_direct)
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_RECT, 128, 8, true>), grid, block8, 0, s, g);
        else
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_RECT, 128, 8, false>), grid, block8, 0, s, g);
        break;
      case GEMM_LOWER:
        if (g_gemm_direct)
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LOWER, 128, 8, true>), grid, block8, 0, s, g);
        else
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LOWER, 128, 8, false>), grid, block8, 0, s, g);
        break;
      case GEMM_LAUUM:
        if (g_gemm_direct)
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LAUUM, 128, 8, true>), grid, block8, 0, s, g);
        else
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LAUUM, 128, 8, false>), grid, block8, 0, s, g);
        break;
      default:
        break;
    }
  } else {
    dim3 grid(ntiles);
    switch (mode) {
      case GEMM_RECT:
        if (g_gemm_direct)
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_RECT, 128, 4, true>), grid, block, 0, s, g);
        else
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_RECT, 128, 4, false>), grid, block, 0, s, g);
        break;
      case GEMM_LOWER:
        if (g_gemm_direct)
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LOWER, 128, 4, true>), grid, block, 0, s, g);
        else
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LOWER, 128, 4, false>), grid, block, 0, s, g);
        break;
      case GEMM_LAUUM:
        if (g_gemm_direct)
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LAUUM, 128, 4, true>), grid, block, 0, s, g);
        else
          hipLaunchKernelGGL((dgemm_nt_kernel<GEMM_LAUUM, 128, 4, false>), grid, block, 0, s, g);
        break;
      default:
        break;
    }
  }
  if (e1) (void)hipEventRecord(e1, s);
}

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
