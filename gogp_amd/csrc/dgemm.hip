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
// both operand tiles (128 rows x 16 doubles = one 128-B line per row) go
// global -> LDS directly (global_load_lds_dwordx4), double-buffered, one barrier per step.
// LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 so
// that the MFMA fragment reads (16 rows x 2 k per 32-lane group, ds_read_b64)
// hit 32 distinct 8-B bank pairs: conflict-free.  The fragment reads are inline-assembly
// ds_read_b64 with explicit lgkmcnt waits, software-pipelined one k-group (4 columns) ahead
// of the MFMAs that consume them (round 3: +2.6 % over the same reads waited for at once).
#include <algorithm>

#include <hip/hip_ext.h>

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
  // GEMM_LAUUM only: the K range of tile (ti,tj) is [ti*BT, kend)
  int kend;
  int trap;  // GEMM_TRAP: skip tiles of strictly upper 256-blocks
  // Tile filter of a sharded (2-D block-cyclic) evaluation, GEMM_RECT only; rule 0: none.
  // The launch covers LOCAL tiles; local tile (ti, tj) lies in the distribution block
  //   global row block  gI = (rblk0 + (ti >> tpb_shift)) * Pr + pr,
  //   global col block  gJ = (cblk0 + (tj >> tpb_shift)) * Pc + pc      (common.h: GemmGrid)
  // rule 1/2: keep the tile iff it belongs to the lower triangle of the GLOBAL matrix (gI > gJ,
  // or gI == gJ and the tile is on/below the diagonal of that block); rule 2 additionally
  // overwrites (beta = 0) the tiles of row block gI == beta0 and accumulates into the others.
  int rule, tpb_shift, rblk0, cblk0, pr, Pr, pc, Pc, beta0;
  int new_row0;  // GEMM_LOWER: tile rows >= new_row0 overwrite C (common.h: GemmGrid); INT_MAX: none
  int ktri;      // GEMM_RECT: B lower triangular, tile column tj sums k < (tj + 1) * BT only
  int prio;      // chain launch: s_setprio 3 (common.h: GemmGrid)
  int krag0;     // RECT / LOWER: tile rows ti >= krag0 start at k = (ti - krag0) * BT (common.h: GemmGrid); INT_MAX: none
  int kbeg0;     // GEMM_LAUUM: second of two launches (common.h: GemmGrid)
  long bstride;  // candidate batching: byte offset of A, B, C per blockIdx.z (common.h: Batch)
#ifdef GOGP_WGSTAMP
  unsigned long long *stamps;  // probe build: this launch's slice of the stamp buffer (common.h), or nullptr
#endif
};

#ifdef GOGP_WGSTAMP
unsigned long long *g_stamp_buf = nullptr;
long long g_stamp_cap = 0, g_stamp_used = 0, g_stamp_nrec = 0;
StampRec g_stamp_rec[1 << 16];
unsigned long long *stamp_reserve(long long nwg, long long tag, hipStream_t s) {
  if (!g_stamp_buf || g_stamp_used + nwg > g_stamp_cap || g_stamp_nrec >= (1 << 16)) return nullptr;
  unsigned long long *p = g_stamp_buf + STAMP_SLOTS * g_stamp_used;
  g_stamp_rec[g_stamp_nrec++] = {g_stamp_used, nwg, tag, (long long)(size_t)s};
  g_stamp_used += nwg;
  return p;
}
#define GOGP_STAMP_WG(k)                                                                                            \
  do {                                                                                                              \
    if (g.stamps && threadIdx.x == 0)                                                                               \
      g.stamps[((long)blockIdx.x + (long)gridDim.x * blockIdx.z) * STAMP_SLOTS + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define GOGP_STAMP_WG(k) \
  do {                   \
  } while (0)
#endif

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
__device__ __forceinline__ void raise_wave_priority() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setprio(3);
#endif
}
// Fragment reads as explicit ds_read_b64.  hipcc fuses the `double` loads of two MFMA tiles into
// ds_read2st64_b64, which the LDS serves in four 16-lane groups over a 32-bank modulus: the rows r and
// r ^ 1 of the XOR swizzle (designed for ds_read_b64: two 32-lane groups, 64 banks) then share a bank,
// SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE, 16 LDS cycles per pair against 2 x 2 (round-3 PMC).
// The compiler does not count an inline-asm read in lgkmcnt, so the waits are explicit too
// (lds_wait<N>) and every fragment is pinned behind its wait by an asm operand.
template <int OFF>
__device__ __forceinline__ double lds_read_b64(unsigned addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
template <int I, int N>
__device__ __forceinline__ void read_frags(double (&f)[N], unsigned addr) {
  if constexpr (I < N) {
    f[I] = lds_read_b64<I * 16 * GEMM_BK * 8>(addr);
    read_frags<I + 1, N>(f, addr);
  }
}
// wait until at most PENDING LDS reads are outstanding, then pin the fragments behind the wait
template <int PENDING, int NA, int NB>
__device__ __forceinline__ void lds_wait(double (&fa)[NA], double (&fb)[NB]) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fa[0]) : "i"(PENDING));
#pragma unroll
  for (int i = 1; i < NA; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
  for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(fb[i]));
}
__device__ __forceinline__ unsigned lds_byte_address(const double *p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const double *)p;
}

__device__ __forceinline__ void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
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
// Operand tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no
// ds_write): a wave's 64 lanes x 16 B land on 1 KB of consecutive LDS = 8 rows of the tile,
// so the XOR swizzle is applied on the GLOBAL side (lane (row, pos) fetches chunk pos ^ swz(row)).
template <int MODE, int BT, int NW>
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
  if (g.prio) raise_wave_priority();
  GOGP_STAMP_WG(0);
#ifdef GOGP_WGSTAMP
  if (g.stamps && threadIdx.x == 0) {
    unsigned long long *st = g.stamps + ((long)blockIdx.x + (long)gridDim.x * blockIdx.z) * STAMP_SLOTS;
    st[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, 32 bits
    st[5] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  }
#endif

  // ---- tile assignment ----------------------------------------------------
  int t = blockIdx.x;
  if (MODE != GEMM_LAUUM && !(MODE == GEMM_RECT && g.rule)) {
    // XCD-aware remap (blocks b and b+8 share an XCD/L2): give each XCD a
    // contiguous chunk of the tile list; bijective for any grid size.
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = t & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
  }
  int ti, tj;
  if (MODE == GEMM_RECT && g.rule) {
    // Filtered launch of the sharded path: the kept tiles form a staircase (global lower
    // triangle), so contiguous chunks per XCD would be badly unbalanced.  Deal the tile ROWS
    // cyclically instead: the XCD group b & 7 takes rows x, x + 8, ... (a row's tiles share the
    // A panel in that XCD's L2).  The grid is 8 * ceil(mt / 8) * nt workgroups.
    const int x = t & 7, slot = t >> 3;
    const int rr = slot / g.nt;
    ti = x + 8 * rr;
    tj = slot - rr * g.nt;
    if (ti >= g.mt) return;
  } else if (MODE == GEMM_RECT) {
    ti = t / g.nt;
    tj = t - ti * g.nt;
    if (g.trap && (tj * BT) / PANEL > (ti * BT) / PANEL) return;  // whole-workgroup exit
  } else {
    ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
  }
  double beta = g.beta;
  if (MODE == GEMM_RECT && g.rule) {  // workgroup-uniform early exit: tiles of the global upper triangle
    const int gI = (g.rblk0 + (ti >> g.tpb_shift)) * g.Pr + g.pr;
    const int gJ = (g.cblk0 + (tj >> g.tpb_shift)) * g.Pc + g.pc;
    const int msk = (1 << g.tpb_shift) - 1;
    if (gI < gJ || (gI == gJ && (ti & msk) < (tj & msk))) return;
    if (g.rule == 2) beta = (gI == g.beta0) ? 0.0 : 1.0;
  }
  if (MODE == GEMM_LOWER && ti >= g.new_row0) beta = 0.0;
  int kbeg = 0, nkt = g.nkt;
  if (MODE == GEMM_RECT && g.ktri) nkt = min(nkt, (tj + 1) * BT / GEMM_BK);
  if (MODE != GEMM_LAUUM && ti > g.krag0) {  // the rows of A below krag0 are zero left of their own diagonal tile
    kbeg = (ti - g.krag0) * BT;
    nkt -= kbeg / GEMM_BK;
  }
  if (MODE == GEMM_LAUUM) {
    kbeg = ti * BT;
    if (kbeg < g.kbeg0) {  // the first launch summed k < kbeg0 into this tile
      kbeg = g.kbeg0;
      beta = 1.0;
    }
    nkt = (g.kend - kbeg) / GEMM_BK;
    if (nkt <= 0) return;  // whole-workgroup exit (tile-uniform)
  }

  const double *Ag = cand(g.A, g.bstride) + (long)ti * BT * g.lda + kbeg;
  const double *Bg = cand(g.B, g.bstride) + (long)tj * BT * g.ldb + kbeg;

  // ---- staging map: thread -> (row, 16-B chunk), NQ rows per operand --------
  const int srow = tid >> 3;  // 0..SROWS-1, +SROWS*q
  const int schunk = tid & 7;
  const int gchunk = schunk ^ ((srow >> 1) & 7);  // SROWS*q never changes swz
  const double *Ap = Ag + (long)srow * g.lda + gchunk * 2;
  const double *Bp = Bg + (long)srow * g.ldb + gchunk * 2;
  const long a_step = (long)SROWS * g.lda, b_step = (long)SROWS * g.ldb;

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
  // LDS byte addresses of the A / B fragment (m = 0 / n = 0) of k-group kk in buffer 0
  unsigned afrag[4], bfrag[4];
  {
    const unsigned lds0 = lds_byte_address(&lds[0][0][0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int xk = (((kk * 2 + fchunk) ^ ((frow >> 1) & 7)) << 1) + fhalf;
      afrag[kk] = lds0 + 8u * (unsigned)(abase + xk);
      bfrag[kk] = lds0 + 8u * (unsigned)(BT * GEMM_BK + bbase + xk);
    }
  }
  constexpr unsigned BUF_BYTES = 2u * BT * GEMM_BK * 8u;  // one double-buffer half: A tile + B tile

  // C fragment of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  double *Cg = cand(g.C, g.bstride) + (long)(ti * BT + wr * WT) * g.ldc + tj * BT + wc * WTN;
  // address = wave-uniform row base (SGPRs) + ONE 32-bit per-lane offset + immediate: per-lane
  // 64-bit row pointers kept across the k loop spill, and a spill reload in the epilogue
  // waits (vmcnt counts stores too) for every C store issued before it
  const int coff = (lane >> 4) * (int)g.ldc + (lane & 15);
  const double alpha = g.alpha;

  // prologue loads of k-tile 0 go out first, the C tile right behind them: the
  // accumulators start at (beta/alpha) C, so the epilogue is stores only and the
  // C read latency overlaps the operand pipeline fill
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    load16_to_lds(Ap + q * a_step, &lds[0][0][(wid * 8 + SROWS * q) * GEMM_BK]);
    load16_to_lds(Bp + q * b_step, &lds[0][1][(wid * 8 + SROWS * q) * GEMM_BK]);
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
  wait_vmcnt0();  // the tile is in LDS
  __syncthreads();
  GOGP_STAMP_WG(1);

  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = (kt + 1 < nkt);
    // fragments of k-group kk + 1 are in flight while the MFMAs of kk issue (two register sets); the
    // first two groups are requested right behind the barrier, the next tile's global -> LDS loads go
    // out behind the first group's MFMAs (issued before the fragment reads instead: K = 16384 72.24
    // against 72.86 TFLOP/s, N = 16384 evaluation +0.3 ms)
    const unsigned curoff = (unsigned)cur * BUF_BYTES;
    double a[2][MT], b[2][NTW];
    read_frags<0, MT>(a[0], afrag[0] + curoff);
    read_frags<0, NTW>(b[0], bfrag[0] + curoff);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) {
        read_frags<0, MT>(a[(kk + 1) & 1], afrag[kk + 1] + curoff);
        read_frags<0, NTW>(b[(kk + 1) & 1], bfrag[kk + 1] + curoff);
        lds_wait<MT + NTW>(a[kk & 1], b[kk & 1]);
      } else {
        lds_wait<0>(a[kk & 1], b[kk & 1]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk & 1][m], b[kk & 1][n], acc[m][n], 0, 0, 0);
      sched_fence();  // or the MFMAs of kk sink below the wait of kk + 1
      if (kk == 0 && more) {
        // buffer cur^1 was last read in step kt-1; every wave has passed that step's barrier
        const double *ap = Ap + (long)(kt + 1) * GEMM_BK;
        const double *bp = Bp + (long)(kt + 1) * GEMM_BK;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          load16_to_lds(ap + q * a_step, &lds[cur ^ 1][0][(wid * 8 + SROWS * q) * GEMM_BK]);
          load16_to_lds(bp + q * b_step, &lds[cur ^ 1][1][(wid * 8 + SROWS * q) * GEMM_BK]);
        }
        sched_fence();
      }
    }
    if (more) wait_vmcnt0();
    __syncthreads();
    cur ^= 1;
  }

  GOGP_STAMP_WG(2);
  // ---- epilogue: stores only
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        (Cg + (long)(m * 16 + 4 * v) * g.ldc)[coff + n * 16] = alpha * acc[m][n][v];
#ifdef GOGP_WGSTAMP
  if (g.stamps) {
    wait_vmcnt0();  // probe build: the stores have left
    GOGP_STAMP_WG(3);
  }
#endif
}

void launch_dgemm_nt(hipStream_t s, GemmMode mode, int mt, int nt, int64_t K, double alpha,
                     const double *A, int64_t lda, const double *B, int64_t ldb,
                     double beta, double *C, int64_t ldc, GemmProfile *prof, const GemmGrid *grid) {
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
  g.kend = (int)K;
  g.rule = 0;
  g.tpb_shift = g.rblk0 = g.cblk0 = g.pr = g.pc = g.beta0 = 0;
  g.Pr = g.Pc = 1;
  g.new_row0 = (mode == GEMM_LOWER && grid && grid->new_row0 >= 0) ? grid->new_row0 : 0x7fffffff;
  g.ktri = (mode == GEMM_RECT && grid && grid->ktri) ? 1 : 0;
  g.prio = grid ? grid->prio : 0;
  g.krag0 = (mode != GEMM_LAUUM && mode != GEMM_TRAP && grid && grid->krag0 >= 0 && !g.ktri) ? grid->krag0 : 0x7fffffff;
  g.kbeg0 = (mode == GEMM_LAUUM && grid) ? grid->kbeg0 : 0;
  g.bstride = tl_batch.stride;
  const unsigned nz = (unsigned)tl_batch.k;
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
  if (mode == GEMM_TRAP) {  // rectangular enumeration, upper 256-blocks skipped in the kernel
    mode = GEMM_RECT;
    g.trap = 1;
    ntiles = mt * nt;
    const int nb = nt / 2;  // 256-blocks across; block column b skips b block rows of 2 x 2 tiles
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
    if (g.rule) {  // count the tiles the filter keeps
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
        flops += 2.0 * (double)(i + 1) * TILE * TILE * (double)(K - std::max<int64_t>((int64_t)i * TILE, g.kbeg0));
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
    prof->flops += flops * nz;
    prof->launches += 1;
    prof->lflops.push_back(flops * nz);
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
      GOGP_KLAUNCH((__VA_ARGS__), (GRID), (BLOCK), 0, s, g);                         \
  } while (0)
  // Small launches (the skinny GEMMs of the panel chain) use 64x64 tiles: 4x the
  // workgroups and a quarter of the per-tile latency.  LAUUM keeps 128 (its K
  // ranges are cut at 128-row granularity).  Launches of >= 3072 tiles and LAUUM use the
  // 8-wave shape of the 128x128 tile (measured 4-9 % faster there), the rest the 4-wave one.
  // (a batched launch counts the tiles of all its candidates: together they fill the chip)
  // ... and so do launches of 513 .. 768 tiles: 128 x 128 tiles have 512 places on the chip (two per CU), so such a
  // launch runs a second, almost empty round at the full per-round price (528 tiles at K = 512, the last fused K^-1
  // update of an N = 4096 evaluation: 203 us); as 64 x 64 tiles it is two rounds of a quarter of the work each
  // (one N = 4096 evaluation 3.54-3.59 -> 3.48-3.50 ms; N = 16384 and 8 candidates at N = 4096 unchanged)
#ifdef GOGP_WGSTAMP
  g.stamps = nullptr;
#endif
  const long total_tiles = (long)ntiles * nz;
  const bool small = (mode != GEMM_LAUUM) && (total_tiles < (grid ? grid->small_below : 384) ||
                                              (total_tiles > 512 && total_tiles <= 768));
  // chain_prio = 1: only the skinny launches (64x64 tiles) raise their priority; 2: every chain launch
  if (g.prio == 1 && !small) g.prio = 0;
  if (small) {
    g.mt = mt * 2;
    g.nt = nt * 2;
    g.tpb_shift += 1;  // distribution blocks counted in 64-wide tiles
    if (g.new_row0 != 0x7fffffff) g.new_row0 *= 2;
    if (g.krag0 != 0x7fffffff) g.krag0 *= 2;  // counted in 64-wide tiles (and 64-column steps of the K start)
    const int n64 = (mode == GEMM_RECT) ? (g.rule ? 8 * ((g.mt + 7) / 8) * g.nt : g.mt * g.nt)
                                        : g.mt * (g.mt + 1) / 2;
#ifdef GOGP_WGSTAMP
    g.stamps = stamp_reserve((long long)n64 * nz, 10000000000LL * 1 + (long long)mode * 100000000LL + (K / 16) * 100000LL + std::min(ntiles, 99999), s);
#endif
    if (mode == GEMM_RECT)
      GOGP_LAUNCH(dim3(n64, 1, nz), dim3(256), dgemm_nt_kernel<GEMM_RECT, 64, 4>);
    else
      GOGP_LAUNCH(dim3(n64, 1, nz), dim3(256), dgemm_nt_kernel<GEMM_LOWER, 64, 4>);
  } else if (mode == GEMM_LAUUM || total_tiles >= 3072) {
    const dim3 grid(g.rule ? 8 * ((mt + 7) / 8) * nt : ntiles, 1, nz), block8(512);
#ifdef GOGP_WGSTAMP
    g.stamps = stamp_reserve((long long)grid.x * nz, 10000000000LL * 3 + (long long)mode * 100000000LL + (K / 16) * 100000LL + std::min(ntiles, 99999), s);
#endif
    if (mode == GEMM_RECT)
      GOGP_LAUNCH(grid, block8, dgemm_nt_kernel<GEMM_RECT, 128, 8>);
    else if (mode == GEMM_LOWER)
      GOGP_LAUNCH(grid, block8, dgemm_nt_kernel<GEMM_LOWER, 128, 8>);
    else
      GOGP_LAUNCH(grid, block8, dgemm_nt_kernel<GEMM_LAUUM, 128, 8>);
  } else {
    const dim3 grid(g.rule ? 8 * ((mt + 7) / 8) * nt : ntiles, 1, nz), block(256);
#ifdef GOGP_WGSTAMP
    g.stamps = stamp_reserve((long long)grid.x * nz, 10000000000LL * 2 + (long long)mode * 100000000LL + (K / 16) * 100000LL + std::min(ntiles, 99999), s);
#endif
    if (mode == GEMM_RECT)
      GOGP_LAUNCH(grid, block, dgemm_nt_kernel<GEMM_RECT, 128, 4>);
    else
      GOGP_LAUNCH(grid, block, dgemm_nt_kernel<GEMM_LOWER, 128, 4>);
  }
#undef GOGP_LAUNCH
}

}  // namespace gogp
