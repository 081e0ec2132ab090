// panel128.hip -- one 128-column step of the blocked Cholesky's dependency chain in ONE launch: the 128 x 128 diagonal
// block is factored AND every row of the panel below it is solved against it (option "chain_split" = 2, api.hip).
//
// Reference counterpart: the diagonal-block factorisation and the panel solve (Dtrsm) of gonum's blocked Dpotrf behind
// mat.Cholesky.Factorize (gp/gp.go:228); "not positive definite" is reported through *info (first failing pivot + 1).
//
// Why: tools/wg_stamps.py (profiles/r05_wg_stamps_*): the chain per 256-panel was diag256 (130 us on ONE compute unit:
// two 128-factorisations, two 128-inverses, four 128^3 products) + a K = 256 panel solve on the tile kernel; option
// "chain_split" moved the products to the tile kernel but still pays a 128 x 128 inverse (22K cycles) and a launch for
// each half's solve.  The inverse exists only to turn the solve into a GEMM.  Here the solve rides on the factorisation
// instead: workgroup g takes the 64 panel rows c1 + 64 g .. and EVERY workgroup factors the diagonal block redundantly
// (the same arithmetic in the same order, so all of them hold the same bits and nothing is exchanged).  The stacked
// matrix M = [D; P] (128 + 64 rows x 128 columns) lives in LDS; per 16-column step the panel rows are rows of M like
// the rows of D below the pivot block: forward-substituted in the lanes of the pivot waves (panel16m) and updated
// by rank-16 MFMA products.  The launch lasts as long as ONE 128-factorisation (the pivot-column chain), whatever the
// number of rows; the 256 x 256 block inverses the substitutions / the triangular inverse / Produce need are formed off
// the chain from the finished factor (diag256.hip: dinv256_blocks).
//
// LDS: D's upper triangle is never touched, so M is stored as eight block columns of 16 columns, block column kb holding
// rows 16 kb .. 191 only, leading dimension 18 doubles (fragment reads of 16 rows x 2 k hit 64 distinct banks):
// 18 * 1088 doubles = 153 KB of the 160 KB a gfx950 compute unit has.
#include <algorithm>

#include "common.h"
#include "pivot16.h"

// The test-hook library compiles this file a second time into its own namespace (-DGOGP_NS=gogp_th
// -DGOGP_BUILD_TESTHOOKS) to get the stamped diagnostic build of the kernel without putting it into libgogp_hip.so.
#ifndef GOGP_NS
#define GOGP_NS gogp
#endif
namespace GOGP_NS {
using namespace gogp;

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int NT = 1024;            // threads per workgroup: sixteen waves (<= 128 VGPRs each) ...
constexpr int NT_MULTI = 512;       // ... eight in the multi-slab build, which keeps more alive through the kernel (<= 256)
constexpr int DR = 128;             // diagonal block
constexpr int PR = 64;              // panel rows per workgroup
constexpr int MR = DR + PR;         // rows of the stacked matrix
constexpr int NRB = MR / 16;        // its 16-row blocks
constexpr int NCB = DR / 16;        // 16-column blocks
constexpr int BLD = 18;             // leading dimension inside a block column
constexpr int MSIZE = BLD * (MR * NCB - 8 * NCB * (NCB - 1));  // 19584 doubles
static_assert(MSIZE * 8 <= 160 * 1024 - 1024, "M must fit the compute unit's LDS");

// first double of block column kb (rows 16 kb .. MR - 1)
__device__ __forceinline__ int cbase(int kb) { return BLD * (MR * kb - 8 * kb * (kb - 1)); }

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Column block kb of M, factored AND solved by one wave in one pass.  Every lane owns TWO rows: row lane & 15 of the
// 16 x 16 pivot block (ad; the block is replicated in the wave's four rows of 16 lanes, so that the pivot column's values
// reach every lane by a DPP broadcast inside its own row) and panel row row0 + lane below it (ap; lanes past row MR - 1
// carry copies of that row).
// Scaling by 1/L_jj and the rank-1 update with the broadcast pivot-column values is exactly the forward substitution of
// the panel row against the block, so the panel solve rides on the factorisation.  Every wave of the step redoes the pivot
// block (the same operations in the same order: the same bits).  The factored pivot block is returned in registers and
// stored by wave 0 after the step's barrier: the other waves read the unfactored block at the start of their own pass.
// (diag256.hip: panel16 does the same with v_readlane broadcasts through SGPRs -- two per value, a third of the pass's
// instructions -- and 48 panel rows per wave in lanes 16..63.)
template <bool SAVE>
__device__ __forceinline__ void panel16m(double *M, int kb, int lane, int row0, bool report, long grow0, long nvalid,
                                         long long *info, double (&ad)[16], double *sv, int svs) {
  const int prow = row0 + lane;
  const bool live = prow < MR;
  const double *srcd = M + cbase(kb) + (lane & 15) * BLD;
  double *srcp = M + cbase(kb) + ((live ? prow : MR - 1) - 16 * kb) * BLD;
  double ap[16];
#pragma unroll
  for (int c = 0; c < 16; c += 2) {
    const f64x2 v = *reinterpret_cast<const f64x2 *>(srcd + c);
    const f64x2 u = *reinterpret_cast<const f64x2 *>(srcp + c);
    ad[c] = v.x;
    ad[c + 1] = v.y;
    ap[c] = u.x;
    ap[c + 1] = u.y;
  }
  int bad = 16;  // first non-positive pivot
  double unused = 0.0;
  // (s_setprio 3 around the pass, measured: no change -- the pass takes 3.5K cycles alone on its SIMD and 5.8K beside
  // three waves of tile updates with or without it: what they share is not issue arbitration)
  PivotColumn<0, false, SAVE>::run(ad, ap, bcast16<0>(ad[0]), 0.0, 0.0, bad, unused, 0, sv, svs);
  // every lane stores: an idle lane carries a copy of row MR - 1, whose owner is in this wave and writes the same bits.
  // (Under `if (live)` hipcc sinks the whole panel-row arithmetic into the branch and keeps 120 broadcasts alive for it.)
#pragma unroll
  for (int c = 0; c < 16; c += 2) *reinterpret_cast<f64x2 *>(srcp + c) = (f64x2){ap[c], ap[c + 1]};
  if (report && bad < 16 && lane == 0 && grow0 + bad < nvalid && *info == 0) *info = (long long)(grow0 + bad + 1);
}

// lanes 0..15 of wave 0: the factored pivot block (upper zeroed), after the step's barrier
__device__ __forceinline__ void store_pivot_block(double *M, int kb, int lane, const double (&keep)[16]) {
  if (lane < 16) {
    double *dst = M + cbase(kb) + lane * BLD;
#pragma unroll
    for (int c = 0; c < 16; c += 2)
      *reinterpret_cast<f64x2 *>(dst + c) = (f64x2){(c <= lane) ? keep[c] : 0.0, (c + 1 <= lane) ? keep[c + 1] : 0.0};
  }
}

// Column block kb of a FURTHER slab of panel rows (P region of M, 64 rows: one per lane) against the pivot block that the
// first slab's pass factored: the saved scalings and the factor block's rows stand in for the pivot chain (pivot16.h:
// SolveColumn) -- the same operations on the panel rows as panel16m does, hence the same bits whatever the number of slabs
// a workgroup takes.
__device__ __forceinline__ void solve16m(double *M, int kb, int lane, const double *sv) {
  const double *srcd = M + cbase(kb) + (lane & 15) * BLD;
  double *srcp = M + cbase(kb) + (DR + lane - 16 * kb) * BLD;
  double ad[16], ap[16];
#pragma unroll
  for (int c = 0; c < 16; c += 2) {
    const f64x2 v = *reinterpret_cast<const f64x2 *>(srcd + c);
    const f64x2 u = *reinterpret_cast<const f64x2 *>(srcp + c);
    ad[c] = v.x;
    ad[c + 1] = v.y;
    ap[c] = u.x;
    ap[c + 1] = u.y;
  }
  SolveColumn<0>::run(ad, ap, sv);
#pragma unroll
  for (int c = 0; c < 16; c += 2) *reinterpret_cast<f64x2 *>(srcp + c) = (f64x2){ap[c], ap[c + 1]};
}

// NTL tiles (it[u], ct[u]) of M -= panel kb's rows of block it[u] times its rows of block ct[u], transposed (rank 16),
// by one wave: every LDS read of all of them is requested before the first MFMA and their accumulator chains interleave
// (one v_mfma_f64_16x16x4_f64 occupies the SIMD's matrix pipe for 64 cycles and a tile is four dependent ones: alone, a
// tile costs three LDS round trips + 256 cycles -- measured 1.1-1.5K cycles per tile, tools/panel_probe.py).
// SPLITK: two accumulators of two MFMAs each per tile (the single tile on the critical path between two pivot blocks).
template <int NTL, bool SPLITK = false>
__device__ __forceinline__ void tile_updates(double *M, const int (&it)[NTL], const int (&ct)[NTL], int kb, int fr, int fk) {
  const int cb = cbase(kb);
  double a[NTL][4], b[NTL][4], cv[NTL][4];
  double *pc[NTL];
#pragma unroll
  for (int u = 0; u < NTL; ++u) {
    const double *pa = M + cb + (16 * (it[u] - kb) + fr) * BLD + fk;
    const double *pb = M + cb + (16 * (ct[u] - kb) + fr) * BLD + fk;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      a[u][k4] = pa[4 * k4];
      b[u][k4] = pb[4 * k4];
    }
  }
#pragma unroll
  for (int u = 0; u < NTL; ++u) {
    pc[u] = M + cbase(ct[u]) + (16 * (it[u] - ct[u]) + fk) * BLD + fr;
#pragma unroll
    for (int v = 0; v < 4; ++v) cv[u][v] = pc[u][4 * v * BLD];
  }
  f64x4 acc[NTL], acc2[NTL];
#pragma unroll
  for (int u = 0; u < NTL; ++u) acc[u] = acc2[u] = (f64x4){0.0, 0.0, 0.0, 0.0};
  if (SPLITK) {
#pragma unroll
    for (int k4 = 0; k4 < 2; ++k4)
#pragma unroll
      for (int u = 0; u < NTL; ++u) {
        acc[u] = mfma(a[u][k4], b[u][k4], acc[u]);
        acc2[u] = mfma(a[u][k4 + 2], b[u][k4 + 2], acc2[u]);
      }
  } else {
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
      for (int u = 0; u < NTL; ++u) acc[u] = mfma(a[u][k4], b[u][k4], acc[u]);
  }
#pragma unroll
  for (int u = 0; u < NTL; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) pc[u][4 * v * BLD] = SPLITK ? cv[u][v] - acc[u][v] - acc2[u][v] : cv[u][v] - acc[u][v];
}

// trailing tile number t of step kb: tiles (i, c), kb + 2 <= c < NCB, c <= i < NRB, column by column
__device__ __forceinline__ void trailing_tile(int t, int kb, int &i, int &c) {
  c = kb + 2;
  while (t >= NRB - c) {
    t -= NRB - c;
    ++c;
  }
  i = c + t;
}

}  // namespace

// A (ld): the 256-block's origin in the matrix being factored (lower triangle valid), Lout (ldl) the same position of the
// factor.  half 0 / 1: columns 0..127 / 128..255 of the block; the diagonal 128-block sits at (off, off), the panel rows
// start 128 below it and there are rows_below of them (a multiple of 64; 0: the last half of the matrix).
// blockIdx.x = g: panel rows 64 g ..; workgroup 0 also writes the diagonal block's factor (and, half 0, the zero block
// right of it) and reports a failing pivot.
// STAMP: diagnostic build (hook library only) that records s_memtime of wave 0 / the last wave of workgroup 0 at the phase
// boundaries of every step into `stamps`.
#define GOGP_PSTAMP(k)                                                                       \
  do {                                                                                       \
    if (STAMP && blockIdx.x == 0 && lane == 0) stamps[k] = __builtin_amdgcn_s_memtime();     \
  } while (0)
// MULTI: a workgroup takes nslab consecutive slabs of 64 panel rows -- the first with the factorisation as above, the others
// by solve16m / the same tile updates on the P region only (launches with more workgroups than the chip has compute units:
// k candidates, or a panel taller than 16384 rows -- every workgroup repeats the diagonal block, so fewer, longer
// workgroups cost less of the GPU's time; one slab each is the latency-optimal form and what MULTI = false compiles to).
template <bool STAMP, bool MULTI>
__global__ __launch_bounds__(MULTI ? NT_MULTI : NT) void panel128_kernel(const double *__restrict__ A, long ld, double *__restrict__ Lout,
                                                       long ldl, int half, long rows_below, long row0, long nvalid,
                                                       long long *info, long bstride, unsigned long long *stamps, int nslab) {
  A = gogp::cand(A, bstride);  // candidate batching (common.h: Batch)
  Lout = gogp::cand(Lout, bstride);
  if (info) info = gogp::cand(info, bstride);
  constexpr int NTK = MULTI ? NT_MULTI : NT, NW = NTK / 64;  // threads / waves of this build
  __shared__ __attribute__((aligned(16))) double M[MSIZE];
  __shared__ __attribute__((aligned(16))) double SV[MULTI ? 2 * DR : 2];  // per pivot column: the two scaling factors
  __shared__ __attribute__((aligned(16))) double SVX[MULTI ? 2 * 64 : 2];  // ... and where the lanes that do not keep them write
  const bool saver = threadIdx.x == 0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the role tests and tile loops stay on the SALU
  const int fr = lane & 15, fk = lane >> 4;
  const long off = half * 128;
  const long slab0 = (long)blockIdx.x * (MULTI ? nslab : 1);  // first slab of 64 panel rows of this workgroup
  const long prow0 = off + DR + slab0 * PR;                   // its first row, from the block's origin
  const bool have_rows = slab0 * PR < rows_below;
  const bool first = blockIdx.x == 0;

  if (w == 0) GOGP_PSTAMP(0);
  // ---- M <- [D; P] ----------------------------------------------------------------------------------------------
  // Every thread owns one pair of columns (2 * lane) of the rows w, w + 16, ..: ALL its requests go out first -- eight
  // of D (lower block columns only) and four of P -- and the LDS stores follow (four requests at a time, as hipcc unrolls
  // the fused loop, the load lasted 10.4K cycles of the kernel's 70K: six HBM / L2 round trips in a row).
  {
    const int cc = lane * 2, cbk = lane >> 3;  // this thread's columns and their block column
    constexpr int QD = DR / NW, QP = PR / NW;
    f64x2 vd[QD], vp[QP];
#pragma unroll
    for (int q = 0; q < QD; ++q) {
      const int i = w + NW * q;
      vd[q] = (f64x2){0.0, 0.0};
      if (cbk <= (i >> 4)) vd[q] = *reinterpret_cast<const f64x2 *>(A + (off + i) * ld + off + cc);
    }
    if (!MULTI) {
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int i = w + NW * q;
        vp[q] = (f64x2){0.0, 0.0};
        if (have_rows) vp[q] = *reinterpret_cast<const f64x2 *>(A + (prow0 + i) * ld + off + cc);
      }
    }
    double *mcol = M + cbase(cbk) - 16 * cbk * BLD + (cc & 15);  // row r of these columns: mcol + r * BLD
#pragma unroll
    for (int q = 0; q < QD; ++q) {
      const int i = w + NW * q;
      if (cbk <= (i >> 4))
        *reinterpret_cast<f64x2 *>(mcol + i * BLD) = (f64x2){(cc <= i) ? vd[q].x : 0.0, (cc + 1 <= i) ? vd[q].y : 0.0};
    }
    if (MULTI) {  // (the multi-slab build keeps more alive through the kernel: its P requests follow D's stores -- 128 VGPRs)
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int i = w + NW * q;
        vp[q] = (f64x2){0.0, 0.0};
        if (have_rows) vp[q] = *reinterpret_cast<const f64x2 *>(A + (prow0 + i) * ld + off + cc);
      }
    }
#pragma unroll
    for (int q = 0; q < QP; ++q) *reinterpret_cast<f64x2 *>(mcol + (DR + w + NW * q) * BLD) = vp[q];
  }
  __syncthreads();

  // ---- blocked right-looking factorisation of M's 128 columns, block size 16 (diag256.hip: potrf128_lds) ----------
  // Per step kb -> kb + 1:
  //   A. column block kb + 1 gets panel kb: tiles (i, kb + 1), i = kb + 1 .. 11, one per wave;
  //   B. waves 0 .. npw - 1 factor + solve column block kb + 1 (panel16m: the pivot block and 64 rows each), the other
  //      waves apply panel kb to the rest of the trailing matrix -- off the critical path.
  double keep[16];
  if (w == 0) GOGP_PSTAMP(1);
  if (w < 3)
    panel16m<MULTI>(M, 0, lane, 16 + 64 * w, first && w == 0, row0 + off, nvalid, info, keep, saver ? SV : SVX + 2 * lane,
                    saver ? 2 : 0);
  if (w == 0) GOGP_PSTAMP(2);
  __syncthreads();
  if (w == 0) GOGP_PSTAMP(3);
#pragma unroll 1
  for (int kb = 0; kb < NCB - 1; ++kb) {
    if (w == 0) store_pivot_block(M, kb, lane, keep);
    for (int i = kb + 1 + w; i < NRB; i += NW) {  // tiles (i, kb + 1), i = kb + 1 .. 11: one per wave (sixteen waves)
      const int it[1] = {i}, ct[1] = {kb + 1};
      tile_updates<1, true>(M, it, ct, kb, fr, fk);
    }
    if (w == 0) GOGP_PSTAMP(8 + kb * 8 + 0);
    __syncthreads();
    if (w == 0) GOGP_PSTAMP(8 + kb * 8 + 1);
    const int rows_b = MR - 16 * (kb + 2);
    const int npw = (rows_b + 63) / 64;
    if (w < npw) {
      panel16m<MULTI>(M, kb + 1, lane, 16 * (kb + 2) + 64 * w, first && w == 0, row0 + off + 16 * (kb + 1), nvalid, info, keep,
                      saver ? SV + 32 * (kb + 1) : SVX + 2 * lane, saver ? 2 : 0);
    } else {
      const int m = NCB - (kb + 2);                        // block columns kb + 2 .. 7
      const int nt3 = m * (NRB - kb - 2) - m * (m - 1) / 2;  // sum over c of (NRB - c)
      const int nwt = NW - npw;
      int t = w - npw;
      for (; t + 2 * nwt < nt3; t += 3 * nwt) {
        int it[3], ct[3];
        trailing_tile(t, kb, it[0], ct[0]);
        trailing_tile(t + nwt, kb, it[1], ct[1]);
        trailing_tile(t + 2 * nwt, kb, it[2], ct[2]);
        tile_updates<3>(M, it, ct, kb, fr, fk);
      }
      if (t + nwt < nt3) {
        int it[2], ct[2];
        trailing_tile(t, kb, it[0], ct[0]);
        trailing_tile(t + nwt, kb, it[1], ct[1]);
        tile_updates<2>(M, it, ct, kb, fr, fk);
      } else if (t < nt3) {
        int it[1], ct[1];
        trailing_tile(t, kb, it[0], ct[0]);
        tile_updates<1>(M, it, ct, kb, fr, fk);
      }
    }
    if (w == 0) GOGP_PSTAMP(8 + kb * 8 + 2);
    if (w == NW - 1) GOGP_PSTAMP(8 + kb * 8 + 3);
    __syncthreads();
    if (w == 0) GOGP_PSTAMP(8 + kb * 8 + 4);
  }
  if (w == 0) store_pivot_block(M, NCB - 1, lane, keep);
  __syncthreads();

  if (w == 0) GOGP_PSTAMP(4);
  // ---- out: the solved panel rows; workgroup 0: the diagonal block's factor ---------------------------------------
  if (have_rows) {
#pragma unroll 4
    for (int idx = tid; idx < PR * 64; idx += NTK) {
      const int i = idx >> 6, cc = (idx & 63) * 2;
      *reinterpret_cast<f64x2 *>(Lout + (prow0 + i) * ldl + off + cc) =
          *reinterpret_cast<const f64x2 *>(M + cbase(cc >> 4) + (DR + i - 16 * (cc >> 4)) * BLD + (cc & 15));
    }
  }
  if (first) {
#pragma unroll 4
    for (int idx = tid; idx < DR * 64; idx += NTK) {
      const int i = idx >> 6, cc = (idx & 63) * 2;
      f64x2 v = {0.0, 0.0};
      if ((cc >> 4) <= (i >> 4))
        v = *reinterpret_cast<const f64x2 *>(M + cbase(cc >> 4) + (i - 16 * (cc >> 4)) * BLD + (cc & 15));
      *reinterpret_cast<f64x2 *>(Lout + (off + i) * ldl + off + cc) = v;
      if (half == 0 && !STAMP)  // upper-right block of the factor's 256-block (the diagnostic build's matrix has none)
        *reinterpret_cast<f64x2 *>(Lout + i * ldl + 128 + cc) = (f64x2){0.0, 0.0};
    }
  }
  // ---- further slabs of this workgroup: M's D region holds the factor, SV the scalings ----------------------------------
  if (MULTI) {
#pragma unroll 1
    for (int sl = 1; sl < nslab; ++sl) {
      const long prow = off + DR + (slab0 + sl) * PR;
      if ((slab0 + sl) * PR >= rows_below) break;
      __syncthreads();  // the previous slab's rows have left M
      {
        const int cc = lane * 2, cbk = lane >> 3;
        constexpr int QP = PR / NW;
        f64x2 vp[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) vp[q] = *reinterpret_cast<const f64x2 *>(A + (prow + w + NW * q) * ld + off + cc);
        double *mcol = M + cbase(cbk) - 16 * cbk * BLD + (cc & 15);
#pragma unroll
        for (int q = 0; q < QP; ++q) *reinterpret_cast<f64x2 *>(mcol + (DR + w + NW * q) * BLD) = vp[q];
      }
      __syncthreads();
      if (w == 0) solve16m(M, 0, lane, SV);
      __syncthreads();
#pragma unroll 1
      for (int kb = 0; kb < NCB - 1; ++kb) {
        // A. the P tiles of column block kb + 1 take panel kb: tiles (8 .. 11, kb + 1), one per wave
        if (w < PR / 16) {
          const int it[1] = {DR / 16 + w}, ct[1] = {kb + 1};
          tile_updates<1, true>(M, it, ct, kb, fr, fk);
        }
        __syncthreads();
        // B. wave 0 solves column block kb + 1 of the slab, the others apply panel kb to the P tiles of the later columns
        if (w == 0) {
          solve16m(M, kb + 1, lane, SV + 32 * (kb + 1));
        } else {
          const int nt3 = (PR / 16) * (NCB - (kb + 2));  // tiles (8 + t % 4, kb + 2 + t / 4)
          for (int t = w - 1; t < nt3; t += NW - 1) {
            const int it[1] = {DR / 16 + (t & 3)}, ct[1] = {kb + 2 + (t >> 2)};
            tile_updates<1>(M, it, ct, kb, fr, fk);
          }
        }
        __syncthreads();
      }
#pragma unroll 4
      for (int idx = tid; idx < PR * 64; idx += NTK) {
        const int i = idx >> 6, cc = (idx & 63) * 2;
        *reinterpret_cast<f64x2 *>(Lout + (prow + i) * ldl + off + cc) =
            *reinterpret_cast<const f64x2 *>(M + cbase(cc >> 4) + (DR + i - 16 * (cc >> 4)) * BLD + (cc & 15));
      }
    }
  }
}

#ifndef GOGP_BUILD_TESTHOOKS
// slabs: 0 = by size -- one slab of 64 rows per workgroup while the launch has at most one workgroup per compute unit (the
// latency-optimal form), more (up to 4) when k candidates or a tall panel ask for more workgroups than that; > 0: forced
// (tests: the result must not depend on it)
void launch_panel128_slabs(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl, int half,
                           int64_t rows_below, int64_t row0, int64_t nvalid, long long *info, int slabs) {
  const unsigned nblk = rows_below > 0 ? (unsigned)(rows_below / PR) : 1u;
  const unsigned k = (unsigned)gogp::tl_batch.k;
  int nslab = slabs;
  if (nslab <= 0) {
    const unsigned total = nblk * k, cus = 256;
    nslab = total <= cus ? 1 : (int)std::min(4u, (total + cus - 1) / cus);
  }
  if (nslab <= 1) {
    GOGP_KLAUNCH((panel128_kernel<false, false>), dim3(nblk, 1, k), dim3(NT), 0, s, A, (long)ld, Lout, (long)ldl, half,
                 (long)rows_below, (long)row0, (long)nvalid, info, gogp::tl_batch.stride, (unsigned long long *)nullptr, 1);
  } else {
    GOGP_KLAUNCH((panel128_kernel<false, true>), dim3((nblk + nslab - 1) / nslab, 1, k), dim3(NT_MULTI), 0, s, A, (long)ld, Lout,
                 (long)ldl, half, (long)rows_below, (long)row0, (long)nvalid, info, gogp::tl_batch.stride,
                 (unsigned long long *)nullptr, nslab);
  }
}
void launch_panel128(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl, int half, int64_t rows_below,
                     int64_t row0, int64_t nvalid, long long *info) {
  launch_panel128_slabs(s, A, ld, Lout, ldl, half, rows_below, row0, nvalid, info, 0);
}
#else
// diagnostic: the stamped build on a device-resident (128 + rows_below) x 128 panel (ld = 128), half 0
void launch_panel128_stamped(hipStream_t s, const double *A, double *Lout, int64_t rows_below, long long *info,
                             unsigned long long *stamps) {
  const unsigned nblk = rows_below > 0 ? (unsigned)(rows_below / PR) : 1u;
  GOGP_KLAUNCH((panel128_kernel<true, false>), dim3(nblk), dim3(NT), 0, s, A, 128L, Lout, 128L, 0, (long)rows_below, 0L,
               (long)(128 + rows_below), info, 0L, stamps, 1);
}
#endif

}  // namespace GOGP_NS
