// diag256.hip -- factor and invert one 256x256 diagonal block in ONE workgroup
// (512 threads = 8 waves: 256-VGPR budget, cheap barriers).  This kernel is the serial link of the blocked
// Cholesky's dependency chain (diag -> panel solve -> next-column update -> diag),
// so it is built for latency: everything O(n^3) inside it runs on
// v_mfma_f64_16x16x4_f64 from LDS, and the only scalar-serial work left is the
// 16x16 base Cholesky (one wave, registers + v_readlane broadcasts).
//
// Reference counterpart: the diagonal-block work of gonum's blocked Dpotrf
// (mat.Cholesky.Factorize, gp/gp.go:228); "not positive definite" (Factorize
// returns false) is reported through *info (first failing pivot + 1).
//
// Block algebra, with A = [[A00, .],[A10, A11]] (lower) and X = L^-1:
//   L00 = chol(A00), X00 = L00^-1
//   L10 = A10 X00^T
//   L11 = chol(A11 - L10 L10^T), X11 = L11^-1
//   X10 = -X11 L10 X00
// Outputs: L (lower, upper zero-filled) and the dense 256x256 inverse Dinv
// (upper zero-filled), which turns every panel solve of the callers into a
// single K=256 GEMM  (panel) * Dinv^T.
//
// chol(128) is blocked by 16 (potrf128_lds): per column block, two waves factor the 16x16
// diagonal block and solve the rows below it in ONE pass (panel16, pivot16.h: one lower row per
// lane beside the replicated block), six waves apply the previous panel to the rest of the
// matrix with MFMA.  The 128x128 inverse is assembled from the eight 16x16 inverses by
// recursive doubling (X21 = -X22 L21 X11 at sizes 16, 32, 64) on MFMA.  The four 128^3
// products (L10, Schur complement, U = X11 L10, X10 = -U X00) skip the zero blocks of their
// triangular operand / the upper tiles of the symmetric result with a balanced tile map
// (wg_gemm128); B is staged once per workgroup through LDS, the Schur product reads both
// operands from S.
//
// Timing (tools/diag_probe.py, cycles of s_memtime; round 4): 2 x 55.6K factor (seven 16-column steps of 6.7-8.4K:
// sixteen dependent pivot columns of ~360 cycles each -- v_rsq_f64, two Newton steps, the scaled pivot row's
// v_readlane, the next pivot's update: ~11 dependent fp64 operations per column -- plus the eight 16x16 inverses),
// 2 x 22K inverse, products 27-34K each incl. epilogues (the MFMA issue alone is ~18K: their eight accumulator chains
// per wave wait for LDS fragments), 19K plain copies = 301K cycles = 128 us per block.
//
// LDS: S[128][130] doubles (padding 2 => the MFMA fragment reads of 16 rows x
// 2 k hit 64 distinct banks), G = 2304 doubles: the eight 16x16 inverses during the factor /
// inverse phases, the double-buffered B chunks during the products.
#include "common.h"
#include "kern_eval.h"
#include "pivot16.h"

// The test-hook library compiles this file a second time into its own namespace
// (-DGOGP_NS=gogp_th -DGOGP_BUILD_TESTHOOKS) to get the stamped diagnostic build of the
// kernel without putting it into libgogp_hip.so.
#ifndef GOGP_NS
#define GOGP_NS gogp
#endif
namespace GOGP_NS {
using namespace gogp;


typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SLD = 130;   // leading dimension of S
constexpr int XLD = 18;    // a 16x16 inverse block: [16][18]
constexpr int GSIZE = 2304;
constexpr int NT = 512;  // threads per workgroup
constexpr int NW = NT / 64;

__device__ __forceinline__ double readlane_d(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Column block kb of the 128x128 matrix in S, factored AND solved by one wave in one pass (pivot16.h): every lane owns
// row lane & 15 of the 16x16 diagonal block (ad; the block is replicated in the wave's four rows of 16 lanes, so the pivot
// column's values reach every lane by a DPP broadcast inside its own row) AND row row0 + lane below it (ap; lanes past row
// 127 carry copies of that row) -- scaling by 1/L_jj and the rank-1 update with the broadcast pivot-column values is
// exactly the forward substitution of the lower row against the block, so the panel solve costs nothing on top of the
// factorisation.  Every wave of the step redoes the diagonal block (the same operations in the same order: the same
// bits).  Writes the solved rows back to S; the not-positive-definite test is branch-free and reported once at the end.
// The factored diagonal block is NOT written here: the other waves of the step read the unfactored block at the start
// of their own pass, so wave 0 keeps its rows (keep[], mine = 1/L_rr of lane r) and stores them with panel16_store_diag
// after the next barrier.  (Until round 5 the broadcasts were v_readlane pairs through SGPRs -- a third of the pass's
// instructions -- 48 lower rows per wave in lanes 16..63, and the update of each column was left to the column that
// needed it: a second dependent chain.  tools/panel_probe.py: 16 columns in ~4.2K cycles against ~5.8K.)
__device__ __forceinline__ void panel16(double *S, int kb, int lane, int row0, bool report,
                                        long grow0, long nvalid, long long *info,
                                        double (&keep)[16], double &keep_rinv) {
  const int r = lane & 15;
  const int prow = row0 + lane;
  const double *srcd = S + (kb * 16 + r) * SLD + kb * 16;
  double *srcp = S + (prow < 128 ? prow : 127) * SLD + kb * 16;
  double ap[16];
#pragma unroll
  for (int c = 0; c < 16; c += 2) {
    const f64x2 v = *reinterpret_cast<const f64x2 *>(srcd + c);
    const f64x2 u = *reinterpret_cast<const f64x2 *>(srcp + c);
    keep[c] = v.x;
    keep[c + 1] = v.y;
    ap[c] = u.x;
    ap[c + 1] = u.y;
  }
  int bad = 16;  // first non-positive pivot
  double mine = 0.0;
  PivotColumn<0, true>::run(keep, ap, bcast16<0>(keep[0]), 0.0, 0.0, bad, mine, r);
  // the values are needed HERE: without this hipcc sinks the lower rows' arithmetic (and 1/L_rr) into the branches that
  // store them, undoing the interleave and keeping every broadcast alive for it (808 VGPR spills)
#pragma unroll
  for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(ap[c]));
  asm volatile("" : "+v"(mine));
  // every lane stores (an idle lane carries a copy of row 127, whose owner is in this wave and writes the same bits) --
  // unless the wave has no lower row at all: row 127 then belongs to the diagonal block itself
  if (row0 < 128) {
#pragma unroll
    for (int c = 0; c < 16; c += 2) *reinterpret_cast<f64x2 *>(srcp + c) = (f64x2){ap[c], ap[c + 1]};
  }
  keep_rinv = mine;
  if (report && bad < 16 && lane == 0 && grow0 + bad < nvalid && *info == 0)
    *info = (long long)(grow0 + bad + 1);
}

// lanes 0..15 of wave 0: factor block (upper zeroed) and 1/L_jj, after the step's barrier
__device__ __forceinline__ void panel16_store_diag(double *S, int kb, double *rinv, int lane,
                                                   const double (&keep)[16], double keep_rinv) {
  if (lane < 16) {
    double *dst = S + (kb * 16 + lane) * SLD + kb * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) dst[c] = (c <= lane) ? keep[c] : 0.0;
    rinv[lane] = keep_rinv;
  }
}

// Inverse of the 16x16 lower-triangular diagonal block kb of S by ONE wave: lane c owns column c of X = L^-1 and runs
// the forward substitution against e_c.  The coefficients L[rr][q] are the same for every lane: each row is ONE
// broadcast read from LDS (every lane the same address: no bank conflict).  (Until round 4 they were v_readlane
// broadcasts of the lanes' own rows: 240 SGPRs at once -- hipcc hoisted them all and spilled every one into VGPR lanes
// and back, 2 x 148 of the kernel's 456 SGPR spills; same operations in the same order: bit-identical results.)
// HAVE_RINV: 1/L_jj is already in rinv[]; otherwise it is formed here.
template <bool HAVE_RINV>
__device__ __forceinline__ void inv16(const double *S, int kb, double *Xb, const double *rinv,
                                      int lane) {
  const int r = lane & 15;
  const double *Lb = S + (kb * 16) * SLD + kb * 16;
  double x[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    double s = (rr == r) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < rr; ++q) s -= Lb[rr * SLD + q] * x[q];
    double ri;
    if (HAVE_RINV) {
      ri = rinv[rr];
    } else {
      const double d = Lb[rr * SLD + rr];
      ri = __builtin_amdgcn_rcp(d);
      ri = fma(ri, fma(-d, ri, 1.0), ri);
      ri = fma(ri, fma(-d, ri, 1.0), ri);
    }
    x[rr] = (rr >= r) ? s * ri : 0.0;
  }
  if (lane < 16) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) Xb[rr * XLD + r] = x[rr];
  }
}

// In-LDS blocked Cholesky of the 128x128 matrix in S (lower triangle valid, upper
// zero), block size 16.  Per step kb -> kb+1:
//   A. the eight waves update column block kb+1 by panel kb (one 16x16 tile each, MFMA);
//   B. waves 0..1 factor + solve column block kb+1 (panel16: the diagonal block and up to
//      2 x 64 rows below, one pass), while waves 2..7 apply panel kb to the rest of the
//      trailing matrix (MFMA rank-16 updates) -- off the critical path.
// The critical path per step is one tile update + one panel16 + two barriers.
// On exit S = L (upper zero) and XD[kb] = inverse of L's kb-th diagonal block (the
// eight inverses are formed at the end by eight waves in parallel).
__device__ __forceinline__ void tile_update16(double *S, int i, int c, int kb, int fr, int fk) {
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k4 = 0; k4 < 4; ++k4) {
    const double av = S[(i * 16 + fr) * SLD + kb * 16 + k4 * 4 + fk];
    const double bv = S[(c * 16 + fr) * SLD + kb * 16 + k4 * 4 + fk];
    acc = mfma(av, bv, acc);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v) S[(i * 16 + fk + 4 * v) * SLD + c * 16 + fr] -= acc[v];
}

// (inlined: as an out-of-line function -- hipcc's choice when left alone -- the call saves four VGPRs to
// scratch, which gives the chain kernel a private segment; A/B: -DGOGP_POTRF_NOINLINE)
#ifdef GOGP_POTRF_NOINLINE
__device__ __noinline__ void potrf128_lds(
#else
__device__ __forceinline__ void potrf128_lds(
#endif
    double *S, double *XD, double *rinv, int tid, long grow0,
                             long nvalid, long long *info, unsigned long long *st = nullptr) {
  const int lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  // rows below block kb: (7-kb)*16; wave w < 2 carries rows (kb+1)*16 + 64w .. +63
  double keep[16], keep_rinv = 0.0;
  if (w < 2) panel16(S, 0, lane, 16 + 64 * w, w == 0, grow0, nvalid, info, keep, keep_rinv);
  if (st && tid == 0) st[1] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  for (int kb = 0; kb < 7; ++kb) {
    // diagonal block kb: every wave of the previous step has read it (barrier above); its
    // next readers are the 16x16 inverses after the loop
    if (w == 0) panel16_store_diag(S, kb, rinv + kb * 16, lane, keep, keep_rinv);
    // ---- A: column block kb+1: tiles (i, kb+1), i = kb+1 .. 7 ---------------------------
    if (kb + 1 + w < 8) tile_update16(S, kb + 1 + w, kb + 1, kb, fr, fk);
    __syncthreads();
    // ---- B -------------------------------------------------------------------------------
    if (w < 2) {
      if ((kb + 2) * 16 + 64 * w < 128 || w == 0)
        panel16(S, kb + 1, lane, (kb + 2) * 16 + 64 * w, w == 0, grow0 + (kb + 1) * 16, nvalid,
                info, keep, keep_rinv);
      if (st && tid == 0 && kb < 2) st[kb * 3 + 2] = __builtin_amdgcn_s_memtime();
    } else {
      // tiles (i, c), kb+2 <= c <= i <= 7, dealt to waves 2..7
      const int m = 6 - kb;  // block columns kb+2 .. 7
      const int nt3 = m * (m + 1) / 2;
      for (int t = w - 2; t < nt3; t += 6) {
        int ii = 0;
        while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
        const int cc = t - ii * (ii + 1) / 2;
        tile_update16(S, kb + 2 + ii, kb + 2 + cc, kb, fr, fk);
      }
    }
    __syncthreads();
  }
  if (w == 0) panel16_store_diag(S, 7, rinv + 7 * 16, lane, keep, keep_rinv);
  __syncthreads();
  // ---- the eight 16x16 inverses, one wave each ----------------------------------------
  inv16<true>(S, w, XD + w * 16 * XLD, rinv + w * 16, lane);
  __syncthreads();
}

// S holds a lower-triangular L (128x128) whose eight 16x16 diagonal-block
// inverses are in XD: overwrite S with X = L^-1 by recursive doubling.
// One product phase of invert128_lds: this wave's (up to) two 16x16 tiles of
// C = A_blk * B_blk, A_blk at (ra, ca), B_blk at (ca, cb) relative to the pair origin.
// All fragments of both tiles are requested first (BS/4 k steps each), then the two
// accumulator chains are interleaved: the LDS latency is paid once per phase instead of
// once per k step, and consecutive MFMAs are independent.
template <int BS>
__device__ __forceinline__ void inv_phase(const double *S, f64x4 (&acc)[2], int w, int fr, int fk,
                                          int aoff, int boff) {
  constexpr int TB = BS / 16, NTPP = TB * TB, NTL = 4 * TB, KS = BS / 4;
  double av[2][KS], bv[2][KS];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int tile = w + u * NW;
    acc[u] = (f64x4){0.0, 0.0, 0.0, 0.0};
    if (tile < NTL) {
      const int p = tile / NTPP, tl = tile - p * NTPP;
      const int ti = tl / TB, tj = tl - ti * TB, o = p * 2 * BS;
      // A_blk row block ti: rows o + BS + ti*16 + fr, columns o + aoff + k
      // B_blk column block tj: rows o + boff + k, columns o + tj*16 + fr
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        av[u][ks] = S[(o + BS + ti * 16 + fr) * SLD + o + aoff + ks * 4 + fk];
        bv[u][ks] = S[(o + boff + ks * 4 + fk) * SLD + o + tj * 16 + fr];
      }
    }
  }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (w + u * NW < NTL) acc[u] = mfma(av[u][ks], bv[u][ks], acc[u]);
}

template <int BS>
__device__ __forceinline__ void inv_store(double *S, const f64x4 (&acc)[2], int w, int fr, int fk,
                                          double sign) {
  constexpr int TB = BS / 16, NTPP = TB * TB, NTL = 4 * TB;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int tile = w + u * NW;
    if (tile < NTL) {
      const int p = tile / NTPP, tl = tile - p * NTPP;
      const int ti = tl / TB, tj = tl - ti * TB, o = p * 2 * BS;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        S[(o + BS + ti * 16 + fk + 4 * v) * SLD + o + tj * 16 + fr] = sign * acc[u][v];
    }
  }
}

template <int BS>
__device__ __forceinline__ void inv_level(double *S, int w, int fr, int fk) {
  f64x4 acc[2];
  // T = B * Xa   (B at (o+BS, o), Xa at (o, o)): A_blk columns o + 0.., B_blk rows o + 0..
  inv_phase<BS>(S, acc, w, fr, fk, 0, 0);
  __syncthreads();
  inv_store<BS>(S, acc, w, fr, fk, 1.0);
  __syncthreads();
  // Z = -Xc * T  (Xc at (o+BS, o+BS)): A_blk columns o + BS.., B_blk rows o + BS..
  inv_phase<BS>(S, acc, w, fr, fk, BS, BS);
  __syncthreads();
  inv_store<BS>(S, acc, w, fr, fk, -1.0);
  __syncthreads();
}

__device__ void invert128_lds(double *S, const double *XD, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  // diagonal blocks <- their inverses
  for (int idx = tid; idx < 8 * 256; idx += NT) {
    const int kb = idx >> 8, e = idx & 255, rr = e >> 4, c = e & 15;
    S[(kb * 16 + rr) * SLD + kb * 16 + c] = XD[kb * 16 * XLD + rr * XLD + c];
  }
  __syncthreads();
  inv_level<16>(S, w, fr, fk);
  inv_level<32>(S, w, fr, fk);
  inv_level<64>(S, w, fr, fk);
}

// C (128x128) = A * B with A = S (LDS, row-major [i][k]) and B from global memory
// (L2-resident: written earlier by this workgroup or by the panel update), staged
// through LDS in 8-deep k chunks, double-buffered in G (free during these phases): every
// B element is fetched ONCE per workgroup, a whole chunk ahead of its use; one barrier per
// chunk.
//
// Every product here has a triangular operand or a symmetric result, so half of the 16x16
// block products are exact zeros (or not needed) and are skipped.  To keep the eight waves
// equally loaded, wave (wr, wc) of the 4x2 arrangement owns the 16-row tiles
// GOGP_RT(0) = wr and GOGP_RT(1) = 7 - wr and the 16-column tiles GOGP_CT(n) = 2n + wc:
//   TRI_A   : A lower triangular  -> row tile rt only needs k < 16 (rt + 1)
//   TRI_B   : B lower triangular  -> column tile ct only needs k >= 16 ct
//   TRI_SYM : C = A A^T, lower tiles only -> tile (rt, ct) needed iff ct <= rt
// (skipped accumulators stay zero).  2 x 128^3 flop on one CU is >= 13.8 us at the MFMA
// rate; this cuts each product to 56-62 % of that.
//   B_NT: B[k][j] = Bg[j*ldb + k]   (rows of Bg are the columns of B)
//         chunk layout Bs[j][9]   : fragment reads hit distinct 8-B banks but one pair
//   else: B[k][j] = Bg[k*ldb + j]
//         chunk layout Bs[k][144] : conflict-free (stride = 16 mod 32 banks)
#define GOGP_RT(m) ((m) ? 7 - wr : wr)
#define GOGP_CT(n) (2 * (n) + wc)
enum { TRI_A = 0, TRI_B = 1, TRI_SYM = 2 };
constexpr int BCH = 8;            // k per chunk
constexpr int BBUF = 1152;        // doubles per chunk buffer (128*9 = 8*144); 2 buffers = GSIZE
static_assert(2 * BBUF <= GSIZE, "chunk buffers must fit G");
template <bool B_NT, int TRI>
__device__ void wg_gemm128(f64x4 (&c)[2][4], const double *S, double *G, const double *Bg, long ldb,
                           int tid) {
  const int lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int wr = w >> 1, wc = w & 1;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) c[m][n] = (f64x4){0.0, 0.0, 0.0, 0.0};
  // staging map: 512 threads x 16 B = one 8 x 128 chunk
  const double *gp;   // this thread's 16 B of chunk 0
  long gstep;         // advance per chunk
  int so;             // LDS offset of its first double
  if (B_NT) {
    const int jj = tid >> 2, kq = (tid & 3) * 2;
    gp = Bg + (long)jj * ldb + kq;
    gstep = BCH;
    so = jj * 9 + kq;
  } else {
    const int kk = tid >> 6, j2 = (tid & 63) * 2;
    gp = Bg + (long)kk * ldb + j2;
    gstep = BCH * ldb;
    so = kk * 144 + j2;
  }
  const double *ap0 = S + (GOGP_RT(0) * 16 + fr) * SLD + fk;
  const double *ap1 = S + (GOGP_RT(1) * 16 + fr) * SLD + fk;
  const int bo = B_NT ? (wc * 16 + fr) * 9 + fk : fk * 144 + wc * 16 + fr;
  constexpr int BN = B_NT ? 32 * 9 : 32;   // fragment offset per n (column tiles 2n + wc)
  constexpr int BS = B_NT ? 4 : 4 * 144;   // ... per k4 step
  // wave-uniform work limits (in chunks / tiles)
  const int e0 = (TRI == TRI_A) ? 2 * (GOGP_RT(0) + 1) : 128 / BCH;  // row tile 0 active for kc < e0
  const int e1 = (TRI == TRI_A) ? 2 * (GOGP_RT(1) + 1) : 128 / BCH;
  const int n0 = (TRI == TRI_SYM) ? (wr >= wc ? (wr - wc) / 2 + 1 : 0) : 4;  // tiles (rt0, ct(n)), n < n0
  const int n1 = (TRI == TRI_SYM) ? (7 - wr - wc) / 2 + 1 : 4;
  if (TRI == TRI_SYM) {
    // C = A A^T with A = S: the B fragments are rows of S as well (B[k][j] = S[j][k]); nothing
    // is staged and nothing synchronised until the end
    const double *bp = S + (wc * 16 + fr) * SLD + fk;  // column tile ct(n): + n * 32 rows
#pragma unroll 2
    for (int k4 = 0; k4 < 32; ++k4) {
      const double a0 = ap0[k4 * 4];
      const double a1 = ap1[k4 * 4];
      double b[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) b[n] = bp[n * 32 * SLD + k4 * 4];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n < n0) c[0][n] = mfma(a0, b[n], c[0][n]);
        if (n < n1) c[1][n] = mfma(a1, b[n], c[1][n]);
      }
    }
    __syncthreads();  // every wave is done reading S
    return;
  }
  // Chunk c travels global -> registers (slot c & 3, requested FOUR chunks before its use: an L2 round trip is about as
  // long as one chunk's MFMAs, so with the request only one chunk ahead -- until round 4 -- every chunk waited for its
  // data: 28-32K cycles per product against 16K of MFMA issue) -> LDS buffer c & 1 (written while chunk c - 1 is
  // multiplied) -> fragments.
  constexpr int NCH = 128 / BCH, PD = 4;
  static_assert(NCH % PD == 0, "the chunk loop is unrolled by the prefetch depth");
  double2 pre[PD];
  pre[0] = *reinterpret_cast<const double2 *>(gp);
#pragma unroll
  for (int u = 1; u < PD; ++u) pre[u] = *reinterpret_cast<const double2 *>(gp + (long)u * gstep);
  G[so] = pre[0].x;
  G[so + 1] = pre[0].y;
  __syncthreads();
#pragma unroll 1
  for (int kc0 = 0; kc0 < NCH; kc0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int kc = kc0 + u;
      const bool more = kc + 1 < NCH;
      // slot u held chunk kc, which went to LDS one step ago
      if (kc + PD < NCH) pre[u] = *reinterpret_cast<const double2 *>(gp + (long)(kc + PD) * gstep);
      const double *bs = G + (kc & 1) * BBUF + bo;
      // first active column tile (TRI_B: ct(n) needs kc >= 2 ct(n))
      int nb = 0;
      if (TRI == TRI_B) {
        const int h = kc / 2 - wc;                 // 2n <= h
        nb = h < 0 ? 0 : min(4, h / 2 + 1);        // column tiles n < nb are active
      }
      const bool act0 = kc < e0, act1 = kc < e1;
      if (act0 || act1) {
#pragma unroll
        for (int k4 = 0; k4 < BCH / 4; ++k4) {
          const double a0 = ap0[kc * BCH + k4 * 4];
          const double a1 = ap1[kc * BCH + k4 * 4];
          double b[4];
#pragma unroll
          for (int n = 0; n < 4; ++n) b[n] = bs[n * BN + k4 * BS];
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const bool on = (TRI == TRI_B) ? n < nb : true;
            if (on && act0 && n < n0) c[0][n] = mfma(a0, b[n], c[0][n]);
            if (on && act1 && n < n1) c[1][n] = mfma(a1, b[n], c[1][n]);
          }
        }
      }
      if (more) {
        double *d = G + ((kc + 1) & 1) * BBUF + so;
        d[0] = pre[(u + 1) % PD].x;
        d[1] = pre[(u + 1) % PD].y;
      }
      __syncthreads();  // chunk kc+1 visible; everybody done with chunk kc (and, at the end, with S)
    }
  }
}

}  // namespace

// A (ld): lower triangle of the 256x256 block (DO_POTRF) or an existing factor
// (!DO_POTRF); Lout (ldl): factor out (DO_POTRF only); Dinv: dense 256x256
// inverse of the factor, leading dimension 256.
// STAMP: diagnostic build that records s_memtime at phase boundaries into `stamps`
// (a buffer of its own; never used by the product path).
#define GOGP_STAMP(k)                                                   \
  do {                                                                  \
    if (STAMP && tid == 0) stamps[k] = __builtin_amdgcn_s_memtime();    \
  } while (0)

template <bool DO_POTRF, bool STAMP, int LDD>
__device__ __forceinline__ void diag256_body(double *S, double *G, double *rinv_s,
                                             const double *__restrict__ A, long ld,
                                             double *__restrict__ Lout, long ldl,
                                             double *__restrict__ Dinv, long row0, long nvalid,
                                             long long *info, unsigned long long *stamps) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int wr = w >> 1, wc = w & 1;
  f64x4 c[2][4];
  // the factor's 10-block as stored in global memory (input when !DO_POTRF)
  const double *L10g = DO_POTRF ? (Lout + 128 * ldl) : (A + 128 * ld);
  const long ld10 = DO_POTRF ? ldl : ld;

#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    const long off = half * 128;
    GOGP_STAMP(half * 8 + 0);
    // ---- S <- diagonal 128-block (half 0: A00; half 1: A11 - L10 L10^T) ----------
    if (half == 0 || !DO_POTRF) {
      // unconditional 16-B loads, then select: a conditional load compiles into a
      // serialised branch + load + vmcnt(0) per element
#pragma unroll 2
      for (int idx = tid; idx < 128 * 64; idx += NT) {
        const int i = idx >> 6, cc = (idx & 63) * 2;
        const f64x2 v = *reinterpret_cast<const f64x2 *>(A + (off + i) * ld + off + cc);
        *reinterpret_cast<f64x2 *>(S + i * SLD + cc) =
            (f64x2){(cc <= i) ? v.x : 0.0, (cc + 1 <= i) ? v.y : 0.0};
      }
      __syncthreads();
    } else {
      // S currently holds L10 (row-major): C = L10 * L10^T, lower tiles only (the
      // accumulators of the tiles above the diagonal stay zero and are masked below)
      wg_gemm128<true, TRI_SYM>(c, S, G, L10g, ld10, tid);
      GOGP_STAMP(26);
      // all 32 A11 requests of a lane go out together (one loop), the LDS stores follow in
      // a second loop: fused, hipcc emits load / vmcnt(0) / store per element (~600 cycles
      // each)
      double a11[2][4][4];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            a11[m][n][v] = A[(128 + GOGP_RT(m) * 16 + fk + 4 * v) * ld + 128 + GOGP_CT(n) * 16 + fr];
      asm volatile("" ::: "memory");  // keep the two loops apart
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int i = GOGP_RT(m) * 16 + fk + 4 * v, cc = GOGP_CT(n) * 16 + fr;
            S[i * SLD + cc] = (cc <= i) ? a11[m][n][v] - c[m][n][v] : 0.0;
          }
      __syncthreads();
    }
    GOGP_STAMP(half * 8 + 1);
    // ---- factor / base inverses ------------------------------------------------------
    if (DO_POTRF) {
      potrf128_lds(S, G, rinv_s, tid, row0 + off, nvalid, info,
                   (STAMP && half == 0) ? stamps + 19 : nullptr);
      GOGP_STAMP(half * 8 + 2);
      // 16-B LDS reads / global stores (S rows are 1040 B apart: 16-B aligned)
#pragma unroll 2
      for (int idx = tid; idx < 128 * 64; idx += NT) {
        const int i = idx >> 6, cc = (idx & 63) * 2;
        *reinterpret_cast<f64x2 *>(Lout + (off + i) * ldl + off + cc) =
            *reinterpret_cast<const f64x2 *>(S + i * SLD + cc);
        if (half == 0)  // upper-right block of the factor
          *reinterpret_cast<f64x2 *>(Lout + i * ldl + 128 + cc) = (f64x2){0.0, 0.0};
      }
    } else {
      inv16<false>(S, w, G + w * 16 * XLD, rinv_s + w * 16, lane);
    }
    __syncthreads();
    GOGP_STAMP(half * 8 + 3);
    // ---- S <- inverse of the 128-block; write it out ----------------------------------
    invert128_lds(S, G, tid);
    GOGP_STAMP(half * 8 + 4);
#pragma unroll 2
    for (int idx = tid; idx < 128 * 64; idx += NT) {
      const int i = idx >> 6, cc = (idx & 63) * 2;
      *reinterpret_cast<f64x2 *>(Dinv + (off + i) * LDD + off + cc) =
          *reinterpret_cast<const f64x2 *>(S + i * SLD + cc);
      if (half == 0) *reinterpret_cast<f64x2 *>(Dinv + i * LDD + 128 + cc) = (f64x2){0.0, 0.0};
    }
    __syncthreads();
    GOGP_STAMP(half * 8 + 5);
    if (half == 0) {
      // ---- L10 = A10 X00^T, computed as C = X00 * A10^T = L10^T ------------------------
      if (DO_POTRF) {
        wg_gemm128<true, TRI_A>(c, S, G, A + 128 * ld, ld, tid);
        GOGP_STAMP(27);
        // S <- L10 (row-major: the transpose of C) for the Schur complement; the factor's
        // 10-block then goes to global memory from S in whole rows (16-B stores) instead of
        // 32 scattered 8-B stores per lane
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int i = GOGP_RT(m) * 16 + fk + 4 * v, j = GOGP_CT(n) * 16 + fr;
              S[j * SLD + i] = c[m][n][v];  // L10[j][i]
            }
        __syncthreads();
#pragma unroll 2
        for (int idx = tid; idx < 128 * 64; idx += NT) {
          const int i = idx >> 6, cc = (idx & 63) * 2;
          *reinterpret_cast<f64x2 *>(Lout + (128 + i) * ldl + cc) =
              *reinterpret_cast<const f64x2 *>(S + i * SLD + cc);
        }
      }
    }
  }
  // ---- X10 = -X11 L10 X00 : S = X11 now ------------------------------------------------
  GOGP_STAMP(16);
  wg_gemm128<false, TRI_A>(c, S, G, L10g, ld10, tid);  // U = X11 * L10
  GOGP_STAMP(17);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        S[(GOGP_RT(m) * 16 + fk + 4 * v) * SLD + GOGP_CT(n) * 16 + fr] = c[m][n][v];
  __syncthreads();
  wg_gemm128<false, TRI_B>(c, S, G, Dinv, LDD, tid);  // U * X00
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        Dinv[(128 + GOGP_RT(m) * 16 + fk + 4 * v) * LDD + GOGP_CT(n) * 16 + fr] = -c[m][n][v];
  GOGP_STAMP(18);
}

// LDD: leading dimension of Dinv (256: a block of its own; 512: a sub-block of the 512x512
// tile inverse of the sharded path)
template <bool DO_POTRF, bool STAMP, int LDD>
__global__ __launch_bounds__(NT) void diag256_kernel(const double *__restrict__ A, long ld,
                                                       double *__restrict__ Lout, long ldl,
                                                       double *__restrict__ Dinv, long row0,
                                                       long nvalid, long long *info,
                                                       unsigned long long *stamps, long bstride) {
  A = gogp::cand(A, bstride);  // candidate batching (common.h: Batch)
  if (Lout) Lout = gogp::cand(Lout, bstride);
  Dinv = gogp::cand(Dinv, bstride);
  if (info) info = gogp::cand(info, bstride);
  __shared__ __attribute__((aligned(16))) double S[128 * SLD];
  __shared__ __attribute__((aligned(16))) double G[GSIZE];
  __shared__ double rinv_s[8 * 16];  // per-wave scratch of base16
#ifdef GOGP_WGSTAMP
  // probe build (common.h): entry / exit on the chip-wide clock; `stamps` doubles as this launch's slice of the buffer
  unsigned long long *wgst = STAMP ? nullptr : stamps;
  if (wgst && threadIdx.x == 0) {
    wgst[blockIdx.z * 8 + 0] = __builtin_amdgcn_s_memrealtime();
    wgst[blockIdx.z * 8 + 4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    wgst[blockIdx.z * 8 + 5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
  }
#endif
  diag256_body<DO_POTRF, STAMP, LDD>(S, G, rinv_s, A, ld, Lout, ldl, Dinv, row0, nvalid, info, stamps);
#ifdef GOGP_WGSTAMP
  if (wgst) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the stores have left
    __syncthreads();
    if (threadIdx.x == 0) wgst[blockIdx.z * 8 + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// ---- one 128 x 128 half of a diagonal block on its own (option "chain_split", api.hip) ---------------------------------
// The 256-block kernel above spends 40 % of its 128 us on four 128^3 products that ONE compute unit has to do alone --
// L10 = A10 X00^T and the Schur complement between the two halves, X10 = -X11 L10 X00 at the end.  Where an evaluation
// is a chain of dependent launches (N <= 8192: sixteen diagonal blocks are 60 % of one N = 4096 evaluation) the sweep
// can instead factor and invert the two 128 x 128 halves with this kernel and give the products to the tile kernel,
// which does them for ALL rows of the panel at once (they ARE the panel solve and the update of the panel's second
// half); X10 is then only needed by the substitutions and the triangular inverse and is formed off the chain.
// half 0: A00 -> L00, X00 (and the zero blocks right of them); half 1: the updated A11 -> L11, X11.
__global__ __launch_bounds__(NT) void diag128_kernel(const double *__restrict__ A, long ld, double *__restrict__ Lout,
                                                      long ldl, double *__restrict__ Dinv, int half, long row0,
                                                      long nvalid, long long *info, long bstride) {
  A = gogp::cand(A, bstride);  // candidate batching (common.h: Batch)
  Lout = gogp::cand(Lout, bstride);
  Dinv = gogp::cand(Dinv, bstride);
  if (info) info = gogp::cand(info, bstride);
  __shared__ __attribute__((aligned(16))) double S[128 * SLD];
  __shared__ __attribute__((aligned(16))) double G[GSIZE];
  __shared__ double rinv_s[8 * 16];
  const int tid = threadIdx.x;
  const long off = half * 128;
#pragma unroll 2
  for (int idx = tid; idx < 128 * 64; idx += NT) {
    const int i = idx >> 6, cc = (idx & 63) * 2;
    const f64x2 v = *reinterpret_cast<const f64x2 *>(A + (off + i) * ld + off + cc);
    *reinterpret_cast<f64x2 *>(S + i * SLD + cc) = (f64x2){(cc <= i) ? v.x : 0.0, (cc + 1 <= i) ? v.y : 0.0};
  }
  __syncthreads();
  potrf128_lds(S, G, rinv_s, tid, row0 + off, nvalid, info);
#pragma unroll 2
  for (int idx = tid; idx < 128 * 64; idx += NT) {
    const int i = idx >> 6, cc = (idx & 63) * 2;
    *reinterpret_cast<f64x2 *>(Lout + (off + i) * ldl + off + cc) = *reinterpret_cast<const f64x2 *>(S + i * SLD + cc);
    if (half == 0) *reinterpret_cast<f64x2 *>(Lout + i * ldl + 128 + cc) = (f64x2){0.0, 0.0};
  }
  __syncthreads();
  invert128_lds(S, G, tid);
#pragma unroll 2
  for (int idx = tid; idx < 128 * 64; idx += NT) {
    const int i = idx >> 6, cc = (idx & 63) * 2;
    *reinterpret_cast<f64x2 *>(Dinv + (off + i) * 256 + off + cc) = *reinterpret_cast<const f64x2 *>(S + i * SLD + cc);
    if (half == 0) *reinterpret_cast<f64x2 *>(Dinv + i * 256 + 128 + cc) = (f64x2){0.0, 0.0};
  }
}

// ---- the dense 256 x 256 inverses of nblk consecutive diagonal blocks of a finished factor, one workgroup each (option
// "chain_split" = 2, api.hip: the chain no longer forms them; the substitutions, the triangular inverse and Produce do need
// them).  blockIdx.x = b: the block at L + b * 256 * (ld + 1), its inverse at Dinv + b * 65536.
__global__ __launch_bounds__(NT) void dinv256_blocks_kernel(const double *__restrict__ L, long ld, double *__restrict__ Dinv,
                                                             long bstride) {
  L = gogp::cand(L, bstride);  // candidate batching (common.h: Batch)
  Dinv = gogp::cand(Dinv, bstride);
  __shared__ __attribute__((aligned(16))) double S[128 * SLD];
  __shared__ __attribute__((aligned(16))) double G[GSIZE];
  __shared__ double rinv_s[8 * 16];
  const long b = blockIdx.x;
  diag256_body<false, false, 256>(S, G, rinv_s, L + b * 256 * (ld + 1), ld, nullptr, 0L, Dinv + b * 65536L, 0L, 0L, nullptr,
                                  nullptr);
}

// ---- tutorial-sized evaluations: N <= 128 observations in ONE workgroup, ONE launch ---------------------------------------
// The reference's own case studies fit 20 ... 44 observations (tutorial/data/*.csv; BASELINE configs[0]: N = 64).  At that
// size the general sweep is fifteen dependent launches of a few microseconds each (kernel trace at N = 64: 0.34 ms per
// Observe, of which the arithmetic is 30 us).  Here workgroup 0 does gp/gp.go:109-236 in one go -- the Gram matrix of
// the n <= 128 observations into LDS (kern_eval.h: simil_value; identity padding as everywhere), its Cholesky factor
// (potrf128_lds), X = L^-1 (invert128_lds), z = X y, alpha = X^T z and, for Observe, K^-1 = X^T X on the matrix cores --
// and writes L, the 256 x 256 block inverse, z, alpha and K^-1 where the general path leaves them; workgroups 1..3 fill
// the constant quadrants of the padded 256-blocks meanwhile.  The log-determinant, the gradient reduction, Produce etc.
// are the general path's kernels on these buffers.  Candidate batching as everywhere (blockIdx.z).
__global__ __launch_bounds__(NT) void tiny_eval_kernel(const DevParams *__restrict__ Pp, const double *__restrict__ X,
                                                        const double *__restrict__ y, long n, double *__restrict__ A,
                                                        double *__restrict__ Lout, double *__restrict__ Dinv,
                                                        double *__restrict__ z, double *__restrict__ alpha,
                                                        long long *info, int want_kinv, long bstride) {
  const DevParams &P = *gogp::cand(Pp, bstride);
  A = gogp::cand(A, bstride);
  Lout = gogp::cand(Lout, bstride);
  Dinv = gogp::cand(Dinv, bstride);
  z = gogp::cand(z, bstride);
  alpha = gogp::cand(alpha, bstride);
  info = gogp::cand(info, bstride);
  const int tid = threadIdx.x;
  if (blockIdx.x != 0) {
    // the constant quadrants of the factor's and the inverse's padded 256-blocks: 1: top right, 2: bottom left (zeros),
    // 3: bottom right (identity: the padding rows' own factor)
    const int q = blockIdx.x, r0 = (q >= 2) ? 128 : 0, c0 = (q & 1) ? 128 : 0;
    for (int idx = tid; idx < 128 * 64; idx += NT) {
      const int i = idx >> 6, cc = (idx & 63) * 2;
      const f64x2 v = {(q == 3 && cc == i) ? 1.0 : 0.0, (q == 3 && cc + 1 == i) ? 1.0 : 0.0};
      *reinterpret_cast<f64x2 *>(Lout + (long)(r0 + i) * 256 + c0 + cc) = v;
      *reinterpret_cast<f64x2 *>(Dinv + (long)(r0 + i) * 256 + c0 + cc) = v;
      // ... and of K^-1's (the input-gradient kernel mirrors and reads the whole padded block)
      if (want_kinv) *reinterpret_cast<f64x2 *>(A + (long)(r0 + i) * 256 + c0 + cc) = v;
    }
    if (q == 3)
      for (int i = tid; i < 128; i += NT) z[128 + i] = alpha[128 + i] = 0.0;
    return;
  }
  __shared__ __attribute__((aligned(16))) double S[128 * SLD];
  __shared__ __attribute__((aligned(16))) double G[GSIZE];
  __shared__ double rinv_s[8 * 16];
  __shared__ double ys[128], zs[128];
  const int D = P.ndim;
  // ---- the Gram matrix, lower triangle, into S (gp/gp.go:109-156; rows / columns >= n: identity) ------------------------
  for (int idx = tid; idx < 128 * 128; idx += NT) {
    const int i = idx >> 7, j = idx & 127;
    double k = 0.0;
    if (j <= i) {
      if (i < n) {
        const double *xi = X + (long)i * D, *xj = X + (long)j * D;
        k = simil_value(P, [&](int d) { return xi[d]; }, [&](int d) { return xj[d]; });
        if (i == j) k += P.noise_var;
      } else {
        k = (i == j) ? 1.0 : 0.0;
      }
    }
    S[i * SLD + j] = k;
  }
  if (tid < 128) ys[tid] = tid < n ? y[tid] : 0.0;
  __syncthreads();
  potrf128_lds(S, G, rinv_s, tid, 0, n, info);
  for (int idx = tid; idx < 128 * 64; idx += NT) {
    const int i = idx >> 6, cc = (idx & 63) * 2;
    *reinterpret_cast<f64x2 *>(Lout + (long)i * 256 + cc) = *reinterpret_cast<const f64x2 *>(S + i * SLD + cc);
  }
  __syncthreads();
  invert128_lds(S, G, tid);  // S = X = L^-1 (lower)
  for (int idx = tid; idx < 128 * 64; idx += NT) {
    const int i = idx >> 6, cc = (idx & 63) * 2;
    *reinterpret_cast<f64x2 *>(Dinv + (long)i * 256 + cc) = *reinterpret_cast<const f64x2 *>(S + i * SLD + cc);
  }
  // ---- z = X y, alpha = X^T z (gp/gp.go:232-236): row tid / 4, a quarter of the sum each, added in a fixed order -------
  {
    const int i = tid >> 2, p = tid & 3;
    double a = 0.0;
    for (int k = p; k <= i; k += 4) a += S[i * SLD + k] * ys[k];
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    if (p == 0) {
      zs[i] = a;
      z[i] = a;
    }
  }
  __syncthreads();
  {
    const int j = tid >> 2, p = tid & 3;
    double a = 0.0;
    for (int k = j + p; k < 128; k += 4) a += S[k * SLD + j] * zs[k];
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 2);
    if (p == 0) alpha[j] = a;
  }
  if (!want_kinv) return;
  // ---- K^-1 = X^T X: S <- X^T in place, then the lower tiles of S S^T on the matrix cores ---------------------------------
  __syncthreads();
  for (int idx = tid; idx < 128 * 128; idx += NT) {
    const int i = idx >> 7, j = idx & 127;
    if (j < i) {
      const double v = S[i * SLD + j];
      S[j * SLD + i] = v;
      S[i * SLD + j] = 0.0;
    }
  }
  __syncthreads();
  f64x4 c[2][4];
  wg_gemm128<true, TRI_SYM>(c, S, G, nullptr, 0, tid);
  const int lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int wr = w >> 1, wc = w & 1;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn)
      if (GOGP_CT(nn) <= GOGP_RT(m)) {
        // both triangles: the input-gradient kernel's mirror step takes the diagonal 32 x 32 tiles as already symmetric
        // (the general path's tile kernel writes its diagonal tiles whole)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const long i = GOGP_RT(m) * 16 + fk + 4 * v, j = GOGP_CT(nn) * 16 + fr;
          A[i * 256 + j] = c[m][nn][v];
          A[j * 256 + i] = c[m][nn][v];
        }
      }
}

#ifndef GOGP_BUILD_TESTHOOKS
void launch_tiny_eval(hipStream_t s, const DevParams *P, const double *X, const double *y, int64_t n, double *A, double *L,
                      double *Dinv, double *z, double *alpha, long long *info, bool want_kinv) {
  GOGP_KLAUNCH(tiny_eval_kernel, dim3(4, 1, (unsigned)gogp::tl_batch.k), dim3(NT), 0, s, P, X, y, (long)n, A, L, Dinv, z, alpha,
               info, want_kinv ? 1 : 0, gogp::tl_batch.stride);
}
#endif

#ifndef GOGP_BUILD_TESTHOOKS
void launch_dinv256_blocks(hipStream_t s, const double *L, int64_t ld, double *Dinv, int nblk) {
  GOGP_KLAUNCH(dinv256_blocks_kernel, dim3((unsigned)nblk, 1, (unsigned)gogp::tl_batch.k), dim3(NT), 0, s, L, (long)ld, Dinv,
               gogp::tl_batch.stride);
}
#endif

#ifndef GOGP_BUILD_TESTHOOKS
void launch_diag128(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl, double *Dinv, int half,
                    int64_t row0, int64_t nvalid, long long *info) {
  GOGP_KLAUNCH(diag128_kernel, dim3(1, 1, (unsigned)gogp::tl_batch.k), dim3(NT), 0, s, A, (long)ld, Lout, (long)ldl, Dinv,
               half, (long)row0, (long)nvalid, info, gogp::tl_batch.stride);
}
#endif

void launch_diag256(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl,
                    double *Dinv, int64_t row0, int64_t nvalid, long long *info) {
  unsigned long long *wgst = nullptr;
#if defined(GOGP_WGSTAMP) && !defined(GOGP_BUILD_TESTHOOKS)
  wgst = gogp::stamp_reserve(gogp::tl_batch.k, 90000000000LL, s);  // tag 9e10: the diagonal-block kernel
#endif
  GOGP_KLAUNCH((diag256_kernel<true, false, 256>), dim3(1, 1, (unsigned)gogp::tl_batch.k), dim3(NT), 0, s,
                     A, (long)ld, Lout, (long)ldl, Dinv, (long)row0, (long)nvalid, info,
                     wgst, gogp::tl_batch.stride);
}

#ifndef GOGP_BUILD_TESTHOOKS
void launch_diag256_ld512(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl,
                          double *Dinv, int64_t row0, int64_t nvalid, long long *info) {
  GOGP_KLAUNCH((diag256_kernel<true, false, 512>), dim3(1), dim3(NT), 0, s, A, (long)ld, Lout,
                     (long)ldl, Dinv, (long)row0, (long)nvalid, info,
                     (unsigned long long *)nullptr, 0L);
}
#endif

void launch_diag256_inv_only(hipStream_t s, const double *L, int64_t ld, double *Dinv) {
  GOGP_KLAUNCH((diag256_kernel<false, false, 256>), dim3(1), dim3(NT), 0, s, L, (long)ld,
                     (double *)nullptr, 0L, Dinv, 0L, 0L, (long long *)nullptr,
                     (unsigned long long *)nullptr, 0L);
}

#ifndef GOGP_BUILD_TESTHOOKS
void launch_diag256_inv_only_ld512(hipStream_t s, const double *L, int64_t ld, double *Dinv) {
  GOGP_KLAUNCH((diag256_kernel<false, false, 512>), dim3(1), dim3(NT), 0, s, L, (long)ld,
                     (double *)nullptr, 0L, Dinv, 0L, 0L, (long long *)nullptr,
                     (unsigned long long *)nullptr, 0L);
}
#endif

#ifdef GOGP_BUILD_TESTHOOKS
// diagnostic: run the stamped build once on a device-resident 256x256 block
void launch_diag256_stamped(hipStream_t s, const double *A, double *Lout, double *Dinv,
                            long long *info, unsigned long long *stamps) {
  GOGP_KLAUNCH((diag256_kernel<true, true, 256>), dim3(1), dim3(NT), 0, s, A, 256L, Lout, 256L,
                     Dinv, 0L, 256L, info, stamps, 0L);
}
#endif

}  // namespace GOGP_NS
