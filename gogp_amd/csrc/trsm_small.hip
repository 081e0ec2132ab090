// trsm_small.hip -- Produce for FEW test points (M <= 64): V = L^-1 Kstar in ONE persistent launch that reads the
// factor once -- the HBM-bound form of gp/gp.go:337-340 (v = K^-1 Kstar by SolveTo) for the reference's own use of
// Produce, one test point per step of its forecast harness (tutorial/tutorial.go:178-179).
//
// The tile-kernel route (api.hip: produce_solve_t) pads the test points to 128 rows and walks ~36 dependent GEMM
// launches of 8-16 workgroups each: 2.0 ms at N = 16384 whatever M <= 64, against 0.13-0.17 ms for one pass over the
// 1.07 GB lower triangle.  Here: blocked forward substitution with the stored 256-block inverses,
//     for B = 0 .. nb-1:   w_B = b_B - sum_{j<B} L[B,j] v_j ;   v_B = Dinv_B w_B,
// left-looking, one workgroup per 64 rows (four per 256-block), ALL in one launch: workgroup t accumulates its rows'
// products with the v_j of earlier blocks as they are published, the four workgroups of a block then exchange their
// parts of w_B and each multiplies its rows of Dinv_B.  Dependencies between workgroups are counters in global
// memory (cdna_hip_programming.md Guideline 16, form R1): payload by write-through (sc1) stores, every storing wave
// drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds to the counter (agent scope); the consumer polls the
// counter with ONE lane (relaxed sc1 loads + s_sleep), a workgroup barrier, then EVERY load of the payload is an sc1
// buffer load to registers (never through this CU's L1).  A workgroup only waits for workgroups of lower blocks or of
// its own block, blocks are dispatched in index order, so the lowest unfinished block is always resident: no
// circular wait; and every spin is bounded -- a workgroup that gives up sets the time-out word, everybody else sees
// it and leaves, the host reports GOGP_EHIP (never a wrong number, never a hung GPU).
//
// Arithmetic: all products on v_mfma_f64_16x16x4_f64 with the right-hand sides as the N dimension (16 or 32 columns;
// M = 1 pays 15 idle columns of a matrix core that has nothing else to do: 55 us of MFMA time at N = 16384).  A wave
// owns 16 rows; lane (row fr, k-quad fk) loads 16 B = L[row][k0 + 2 fk .. +1] -- 64 contiguous bytes per row and load,
// the row's 128-B line used by two consecutive loads -- and feeds the two doubles to two MFMAs whose B operands are
// v[k0 + 2 fk][j] and v[k0 + 2 fk + 1][j]: the k order inside a product is free as long as both operands agree.
// v_j lives in LDS in the same pair-interleaved layout it has in global memory ([k / 2][j][k % 2]): one
// ds_read_b128 per MFMA pair, conflict-free (16 lanes x 16 B contiguous), staged by a straight 16-B copy.
// The loads of L run four 64-column chunks (one whole 256-block) ahead of the MFMAs in a register ring and do not
// wait for any counter (L is constant): a workgroup behind the frontier streams at the rate HBM gives it, and a
// workgroup AT the frontier has its operands in registers when v_j arrives.
#include "common.h"

namespace gogp {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int TS_RT = 64;                  // rows per workgroup (4 waves x 16)
constexpr int TS_WPB = PANEL / TS_RT;      // workgroups per 256-block
constexpr unsigned TS_SPIN_MAX = 1u << 22;  // polls (~1 us each with the s_sleep) before a workgroup gives up

// TL: element type of the matrices (the factor, its block inverses, the right-hand sides): double, or float on the fp32
// path -- widened as they arrive in registers; the substitution itself (v, w, every sum) is fp64 either way
template <class TL>
struct TsArgs {
  const TL *L;
  long ld;
  const TL *Dinv;  // nb blocks of 256 x 256 (lower, upper zero)
  const TL *KsT;   // right-hand sides, [j][i] (row j = test point), leading dimension ldk
  long ldk;
  int j0;              // first right-hand side of this launch
  double *Vk;          // solution, pair-interleaved: element (row k, rhs j) at ((k >> 1) * MP + j) * 2 + (k & 1)
  double *Wk;          // w_B of the block steps, same layout
  unsigned *cnt;       // [nb] arrivals of w parts
  unsigned *done;      // [nb] arrivals of v parts
  unsigned *tmo;       // time-out word (0: fine)
  double *sqpart;      // [nwg][MP] partial sums of squares
  int nb;
};

__device__ __forceinline__ unsigned ld_u32_agent(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dword sc1
}

// one lane: spin until *p >= target.  false: somebody (maybe this lane) gave up.
__device__ __forceinline__ bool wait_count(const unsigned *p, unsigned target, unsigned *tmo, unsigned code) {
  for (unsigned spins = 0;; ++spins) {
    if (ld_u32_agent(p) >= target) return true;
    if ((spins & 31u) == 31u && ld_u32_agent(tmo) != 0u) return false;
    if (spins >= TS_SPIN_MAX) {
      __hip_atomic_store(tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    if (spins >= 256u) __builtin_amdgcn_s_sleep(4);  // a workgroup at the frontier polls back to back (the hop is on the critical path)
  }
}

__device__ __forceinline__ f64x4 mfma4(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

}  // namespace

// NT: 16-column tiles of right-hand sides the MFMAs work on (MP = 16 NT columns per launch).  MC: columns that travel
// between workgroups -- MC = MP: the blocks of v / w are exchanged in the MFMA operand layout ([k / 2][MP][2], staged
// by a straight 16-B copy); MC = 1, 2, 4, 8 (NT = 1): only the MC live columns travel ([k][MC]; one row per thread of
// the staging copy), the other columns of the LDS operand are zeros written once -- one test point hands 2 KB from
// block to block instead of 32 KB, and every hand-off is on the substitution's critical path.
template <int NT, int MC, class TL>
__global__ __launch_bounds__(256, 2) void trsm_small_kernel(TsArgs<TL> g) {
  constexpr int MP = 16 * NT;
  constexpr bool FULL = MC == MP;
  static_assert(FULL || (NT == 1 && (MC == 1 || MC == 2 || MC == 4 || MC == 8)), "compact exchange: NT = 1, MC in {1, 2, 4, 8}");
  constexpr int VQ = PANEL * MP / 2 / 256;  // 16-B pieces of one block of v per thread: 8 NT
  // one block of v / w: [128][MP][2], and behind it the poll's verdict (no static LDS in front of the dynamic
  // region: a 4-byte static would leave its base off the 16-B alignment the ds_read_b128 below need)
  extern __shared__ __attribute__((aligned(16))) double Vs[];
  volatile int &ok_s = *reinterpret_cast<volatile int *>(Vs + PANEL * MP);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int wg = blockIdx.x;
  const int B = wg / TS_WPB, sub = wg - B * TS_WPB;
  const long r0 = (long)wg * TS_RT + 16 * w;  // first row of this wave
  // 64-column chunks of this workgroup: L[rows, 0:256 B] (4 B of them), then ALL FOUR of Dinv_B[rows, 0:256]: the chunks
  // right of the diagonal (c > sub) are zeros and add nothing, but every ring slot stays in use to the end -- a slot that
  // is never read again is a register hipcc hands to other code, which then waits for the load still in flight into it

  // ---- operand ring: chunk q -> this lane's eight 16-B pieces --------------------------------------
  const TL *Lrow = g.L + (r0 + fr) * g.ld + 2 * fk;
  const TL *Drow = g.Dinv + (long)B * PANEL * PANEL + (long)(sub * TS_RT + 16 * w + fr) * PANEL + 2 * fk;
  // ---- accumulators: acc = sum_j L v_j - b  (so w = -acc); loaded BEFORE the ring's first requests (vector loads
  // return in order: every later wait for a ring slot then covers them, see trsv_granule_kernel) ----------------
  f64x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      acc[nt][v] = -(double)g.KsT[(long)(g.j0 + fr + 16 * nt) * g.ldk + r0 + fk + 4 * v];
  f64x2 ar[4][8];
  auto issue = [&](int q, f64x2(&slot)[8]) {
    const TL *p = (q < 4 * B) ? Lrow + (long)q * 64 : Drow + (long)(q - 4 * B) * 64;
    if constexpr (sizeof(TL) == 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) slot[i] = *reinterpret_cast<const f64x2 *>(p + 8 * i);
    } else {  // float matrices: the same two elements per lane and piece, 8 bytes, widened on arrival
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float2 v = *reinterpret_cast<const float2 *>(p + 8 * i);
        slot[i] = (f64x2){(double)v.x, (double)v.y};
      }
    }
  };
  // (chunk indices are clamped to the last one instead of guarded: a conditional load makes every ring register a
  // phi of "old or new" and hipcc then keeps both -- 137 VGPR spills in the one-right-hand-side kernel; the repeated
  // load of the last chunk lands in a slot nobody reads any more)
  issue(0, ar[0]);  // chunks 0 .. 3: block 0 of L, or (B = 0) this workgroup's rows of Dinv_0 themselves
  issue(1, ar[1]);
  issue(2, ar[2]);
  issue(3, ar[3]);


  const auto vsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g.Vk, 0, (int)((long)g.nb * PANEL * MC * 8), 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g.Wk, 0, (int)((long)g.nb * PANEL * MC * 8), 0x00020000);
  if (!FULL) {  // the columns that never travel: zero, once (the first barrier below separates this from the staging)
    for (int idx = tid; idx < PANEL * MP; idx += 256) Vs[idx] = 0.0;
  }

  // one 256-block of v (from Vk) or w (from Wk) -> LDS; every load sc1 (the bytes were written by other workgroups
  // of this launch), all of a thread's loads in flight before the first LDS store
  auto stage = [&](bool from_w, int blk) {
    if constexpr (FULL) {
      u32x4 t[VQ];
      const unsigned base = (unsigned)blk * (unsigned)(PANEL * MP * 8);
#pragma unroll
      for (int q = 0; q < VQ; ++q)
        t[q] = from_w ? __builtin_amdgcn_raw_buffer_load_b128(wsrc, base + (unsigned)(tid + 256 * q) * 16u, 0, 16)
                      : __builtin_amdgcn_raw_buffer_load_b128(vsrc, base + (unsigned)(tid + 256 * q) * 16u, 0, 16);
#pragma unroll
      for (int q = 0; q < VQ; ++q) *reinterpret_cast<u32x4 *>(Vs + 2 * (tid + 256 * q)) = t[q];
    } else {
      // thread t <-> row t of the block: its MC doubles, scattered into the operand layout
      const unsigned off = ((unsigned)blk * PANEL + (unsigned)tid) * (unsigned)(MC * 8);
      double x[MC];
      if constexpr (MC == 1) {
        const u32x2 t = from_w ? __builtin_amdgcn_raw_buffer_load_b64(wsrc, off, 0, 16)
                               : __builtin_amdgcn_raw_buffer_load_b64(vsrc, off, 0, 16);
        x[0] = __hiloint2double((int)t.y, (int)t.x);
      } else {
        u32x4 t[MC / 2];
#pragma unroll
        for (int q = 0; q < MC / 2; ++q)
          t[q] = from_w ? __builtin_amdgcn_raw_buffer_load_b128(wsrc, off + 16u * q, 0, 16)
                        : __builtin_amdgcn_raw_buffer_load_b128(vsrc, off + 16u * q, 0, 16);
#pragma unroll
        for (int q = 0; q < MC / 2; ++q) {
          x[2 * q] = __hiloint2double((int)t[q].y, (int)t[q].x);
          x[2 * q + 1] = __hiloint2double((int)t[q].w, (int)t[q].z);
        }
      }
#pragma unroll
      for (int j = 0; j < MC; ++j) Vs[2 * ((tid >> 1) * MP + j) + (tid & 1)] = x[j];
    }
  };
  // the MFMAs of one 64-column chunk c (0..3) of the staged block
  auto chunk_mma = [&](int c, const f64x2(&slot)[8], f64x4(&a)[NT]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f64x2 b = *reinterpret_cast<const f64x2 *>(Vs + 2 * ((c * 32 + i * 4 + fk) * MP + fr + 16 * nt));
        a[nt] = mfma4(slot[i].x, b.x, a[nt]);
        a[nt] = mfma4(slot[i].y, b.y, a[nt]);
      }
    }
  };
  // store this wave's 16 x MP accumulator tile to row block (rows r0 ..) of dst, write-through
  auto publish = [&](double *dst, const f64x4(&a)[NT], double sign) {
    if constexpr (FULL) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const long k = r0 + fk + 4 * v;
          __hip_atomic_store(dst + ((k >> 1) * MP + fr + 16 * nt) * 2 + (k & 1), sign * a[nt][v], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 sc1
        }
    } else {
      if (fr < MC) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          __hip_atomic_store(dst + (r0 + fk + 4 * v) * MC + fr, sign * a[0][v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the barrier / the counter add
  };

  // ---- blocks j < B: acc += L[rows, block j] v_j --------------------------------------------------------
  // A block's four chunks are multiplied FIRST and the next block's 32 loads per wave requested behind them (the ring
  // is exactly one block: all four slots are free then).  Interleaved, every load instruction waited for room in the
  // compute unit's memory pipeline (~40 GB/s per unit) with the next chunk's MFMAs behind it in program order: 3.6 us
  // per block (measured on the one-right-hand-side kernel by leaving the loads out).  The last block requests nothing
  // before w is published: the drain of the write-through stores would wait for those loads too.
  auto issue_block = [&](int qb) {
#pragma unroll
    for (int c = 0; c < 4; ++c) issue(qb + c, ar[c]);
  };
  auto block_step = [&](int j) -> bool {
    if (tid == 0) ok_s = wait_count(g.done + j, TS_WPB, g.tmo, 0x100u + (unsigned)wg) ? 1 : 0;
    __syncthreads();  // the poll has matched; everybody is done with the previous block in LDS
    if (!ok_s) return false;
    stage(false, j);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) chunk_mma(c, ar[c], acc);
    return true;
  };
  for (int j = 0; j < B - 1; ++j) {
    if (!block_step(j)) return;
    issue_block(4 * (j + 1));
  }
  if (B > 0 && !block_step(B - 1)) return;
  // ---- diagonal step: publish w, wait for the block's other parts, v_rows = Dinv_B[rows, :] w_B -----------
  publish(g.Wk, acc, -1.0);
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(g.cnt + B, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (B > 0) issue_block(4 * B);  // this workgroup's rows of Dinv_B: they land while the block's w parts travel
  if (tid == 0) ok_s = wait_count(g.cnt + B, TS_WPB, g.tmo, 0x200u + (unsigned)wg) ? 1 : 0;
  __syncthreads();
  if (!ok_s) return;
  stage(true, B);
  __syncthreads();
  f64x4 res[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) res[nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
  // the four chunks of Dinv's rows (lower triangular: zeros right of the diagonal); the ring slot of chunk 4 B + c is c
#pragma unroll
  for (int c = 0; c < 4; ++c) chunk_mma(c, ar[c], res);
  publish(g.Vk, res, 1.0);
  // partial sums of squares of this workgroup's rows, fixed order
  double sq[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    double s = (res[nt][0] * res[nt][0] + res[nt][1] * res[nt][1]) + (res[nt][2] * res[nt][2] + res[nt][3] * res[nt][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    sq[nt] = s;
  }
  __syncthreads();  // every wave has drained its stores of v; LDS is free
  if (tid == 0) __hip_atomic_fetch_add(g.done + B, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane < 16)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Vs[w * MP + lane + 16 * nt] = sq[nt];
  __syncthreads();
  if (tid < MP) g.sqpart[(long)wg * MP + tid] = (Vs[tid] + Vs[MP + tid]) + (Vs[2 * MP + tid] + Vs[3 * MP + tid]);
}

// dq[j0 + j] = sum over the workgroups of sqpart[wg][j], in workgroup order; tmo_out = the time-out word
__global__ void trsm_small_finish_kernel(const double *__restrict__ sqpart, int nwg, int MP, int m, int j0,
                                         double *__restrict__ dq) {
  const int j = threadIdx.x;
  if (j >= MP || j0 + j >= m) return;
  double s = 0.0;
  for (int q = 0; q < nwg; ++q) s += sqpart[(long)q * MP + j];
  dq[j0 + j] = s;
}

// ---- ONE right-hand side (the reference's forecast harness: tutorial/tutorial.go:178-179) ---------------------------
// The kernel above needs seven memory round trips per 256-block on the substitution's critical path (counter add, poll,
// staging load, and the drain of the write-through stores, twice over): ~10 us per block, 0.8 ms at N = 16384 whatever the
// payload.  With ONE right-hand side a block of v is 2 KB and can travel as DATA-TAGGED GRANULES (Guideline 16, form R2):
// every double is one 16-byte {tag, low word, tag, high word} written by ONE write-through store; the consumer's thread
// t re-reads granule t of the block (sc1 loads) until both tags carry the call's epoch -- the data is the flag: no
// counter, no drain, no separate staging load; a hop is one store and one (repeated) load.  The arithmetic moves to the
// vector ALU (a 16-column MFMA tile would carry 15 zeros): lane (row fr, k-quad fk) multiplies its 16 B of L with the
// matching pair of v from LDS (a 4-address broadcast read) and the four k-quads of a row are added once per phase.
// Same geometry, ring and time-out discipline as above; sums in a fixed order.
struct T1Args {
  const double *L;
  long ld;
  const double *Dinv;
  const double *b;  // the right-hand side (row 0 of KsT)
  u32x4 *Vg;        // [npad] granule pairs of v; zeroed before the launch (tag 0 = not written)
  u32x4 *Wg;        // [npad] granule pairs of w
  unsigned *tmo;
  double *sqpart;   // [nwg]
  int nb;
  unsigned long long *stamps;  // diagnostics (tools/trsv_stamps.py): 8 s_memrealtime stamps per workgroup, or nullptr
};
// set by the stamp probe only (tools/trsv_stamps.py, through the mangled name); never by the product path
unsigned long long *g_ts_stamps = nullptr;
#define TS_STAMP(k)                                                                             \
  do {                                                                                          \
    if (g.stamps && lane == 0) g.stamps[(long)wg * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)

// Five waves: waves 0..3 stream L and do the arithmetic (16 rows each), wave 4 is the COURIER -- it issues no other
// memory instruction than the sweeps and runs one block ahead: v_{j+1} is swept into the other half of a double buffer
// while the streaming waves multiply v_j.
// What a block step costs was measured piece by piece (tools/trsv_stamps.py, and builds with single pieces left out): the
// two hops 0.85 + 1.3 us, the arithmetic 0.3 + 0.3 us -- and 3.6 us for ISSUING the next block's 32 loads per wave between
// the multiply-adds: one compute unit takes in ~40 GB/s, a load instruction does not issue before the memory pipeline has
// room, and the wave's next multiply-add waits behind it in program order.  So (1) a block's four chunks are multiplied
// FIRST and the next block's loads requested behind them (the ring is exactly one block: all four slots are free then),
// the last block requests nothing; (2) this workgroup's rows of Dinv_B do not go through the ring at all: each streaming
// wave sends its 16 rows x 64 (sub + 1) columns global -> LDS (LDS-DMA, no registers) before anything else, in the very
// lane order the diagonal step reads them back.  128 KB of LDS: one workgroup per compute unit (N = 16384: 256 workgroups).
__device__ __forceinline__ void ts_load16_to_lds(const double *gsrc, double *lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gsrc, lds_wave_base, 16, 0, 0);
#endif
}
constexpr int TS_DINV_LDS = 4 * 4 * 8 * 128;  // doubles: [wave][chunk][piece] images of 64 lanes x 16 B
__global__ __launch_bounds__(320) void trsv_granule_kernel(T1Args g) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];
  double *dl = dyn;                                         // Dinv rows, TS_DINV_LDS doubles
  double(*vs)[PANEL] = reinterpret_cast<double(*)[PANEL]>(dyn + TS_DINV_LDS);  // [2][256]
  double *red = dyn + TS_DINV_LDS + 2 * PANEL;              // [4]
  volatile int &fail_s = *reinterpret_cast<volatile int *>(red + 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool courier = w == 4;
  const int fr = lane & 15, fk = lane >> 4;
  const int wg = blockIdx.x;
  const int B = wg / TS_WPB, sub = wg - B * TS_WPB;
  const long r0 = (long)wg * TS_RT + 16 * (courier ? 0 : w);
  const double *Lrow = g.L + (r0 + fr) * g.ld + 2 * fk;
  const double *Drow = g.Dinv + (long)B * PANEL * PANEL + (long)(sub * TS_RT + 16 * (courier ? 0 : w) + fr) * PANEL + 2 * fk;
  const auto vsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g.Vg, 0, (int)((long)g.nb * PANEL * 16), 0x00020000);
  const auto wsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g.Wg, 0, (int)((long)g.nb * PANEL * 16), 0x00020000);
  if (tid == 0) fail_s = 0;
  __syncthreads();

  // the courier's sweep: lane l re-reads the granule pairs l, l + 64, l + 128, l + 192 of block blk until all 256 carry
  // the tag, then writes the doubles to buf
  auto sweep = [&](bool from_w, int blk, double *buf, unsigned code) {
    const int slack = from_w ? 0 : min(B - 1 - blk, 16);  // block steps until this workgroup is the frontier
    const unsigned off = ((unsigned)blk * PANEL + (unsigned)lane) * 16u;
    u32x4 x[4];
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        x[i] = from_w ? __builtin_amdgcn_raw_buffer_load_b128(wsrc, off + 1024u * i, 0, 16)
                      : __builtin_amdgcn_raw_buffer_load_b128(vsrc, off + 1024u * i, 0, 16);
      bool ok = true;
#pragma unroll
      for (int i = 0; i < 4; ++i) ok = ok && x[i].x == 1u && x[i].z == 1u;
      if (__all(ok)) break;
      bool give_up = spins >= TS_SPIN_MAX;
      if ((spins & 31u) == 31u) give_up = give_up || ld_u32_agent(g.tmo) != 0u;
      if (give_up) {  // wave-uniform
        if (lane == 0) {
          __hip_atomic_store(g.tmo, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fail_s = 1;
        }
        break;
      }
      // Every workgroup behind block blk needs these 4 KB next, and they all finished the previous block at about the
      // same time: 250 couriers re-reading the SAME 64 lines back to back are a hot spot on one memory channel -- in
      // front of the granule stores they are waiting for.  Only the block that is next in line (slack 0) polls back to
      // back; the others can afford to see v_j late by their distance from the frontier and sleep in proportion.
      for (int z = 0; z < slack; ++z) __builtin_amdgcn_s_sleep(8);
      if (spins >= 64u) __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) buf[lane + 64 * i] = __hiloint2double((int)x[i].w, (int)x[i].y);
  };

  if (courier) {
    // barrier sequence (every wave passes the same ones): one per L block ("v_j is in LDS"), then "w published",
    // "w_B is in LDS", and the final one of the reduction
    TS_STAMP(0);
    for (int j = 0; j < B; ++j) {
      sweep(false, j, vs[j & 1], 0x100u + (unsigned)wg);
      if (j == B - 1) TS_STAMP(1);  // the last v_j has arrived
      __syncthreads();  // v_j ready; the streaming waves are done with v_{j-1}: its buffer is free for v_{j+1}
      if (fail_s) return;
    }
    __syncthreads();  // the streaming waves have published their parts of w_B
    TS_STAMP(3);
    sweep(true, B, vs[B & 1], 0x200u + (unsigned)wg);
    TS_STAMP(4);
    __syncthreads();
    if (fail_s) return;
    __syncthreads();
    return;
  }

  // ---- streaming waves ----------------------------------------------------------------------------------------------
  // the right-hand side FIRST (vector loads return in order: every later wait also covers it), then this wave's rows of
  // Dinv_B straight into LDS: piece (c, i) = 16 rows x 64 B, the lanes in the order the diagonal step reads them back
  const double brow = g.b[r0 + fr];
  double *dlw = dl + w * (4 * 8 * 128);
  for (int c = 0; c <= sub; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) ts_load16_to_lds(Drow + c * 64 + 8 * i, dlw + (c * 8 + i) * 128);
  f64x2 ar[4][8];
  auto issue_block = [&](int j) {  // the four chunks of L[rows, block j] into the four ring slots
    const double *p = Lrow + (long)j * PANEL;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 8; ++i) ar[c][i] = *reinterpret_cast<const f64x2 *>(p + c * 64 + 8 * i);
  };
  // this wave's 16 rows of one 64-column chunk c of the staged block: lane (fr, fk) adds its 16 of the 64 columns
  // (the chunk's eight LDS reads first, then two independent chains of eight multiply-adds)
  auto chunk_fma = [&](const double *buf, int c, const f64x2(&slot)[8], double &acc) {
    f64x2 vv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) vv[i] = *reinterpret_cast<const f64x2 *>(buf + c * 64 + i * 8 + 2 * fk);
    double e = 0.0, o = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      e = fma(slot[i].x, vv[i].x, e);
      o = fma(slot[i].y, vv[i].y, o);
    }
    acc += e + o;
  };
  // one write-through 16-byte store per row: {tag, low, tag, high}
  auto publish = [&](bool to_w, double val) {
    if (fk == 0) {
      u32x4 gr;
      gr.x = 1u;
      gr.y = (unsigned)__double2loint(val);
      gr.z = 1u;
      gr.w = (unsigned)__double2hiint(val);
      if (to_w)
        __builtin_amdgcn_raw_buffer_store_b128(gr, wsrc, (unsigned)(r0 + fr) * 16u, 0, 16);
      else
        __builtin_amdgcn_raw_buffer_store_b128(gr, vsrc, (unsigned)(r0 + fr) * 16u, 0, 16);
    }
  };

  double acc = 0.0;
  if (B > 0) {
    issue_block(0);
    for (int j = 0; j < B - 1; ++j) {
      __syncthreads();  // v_j is in vs[j & 1]
      if (fail_s) return;
      const double *buf = vs[j & 1];
#pragma unroll
      for (int c = 0; c < 4; ++c) chunk_fma(buf, c, ar[c], acc);
      issue_block(j + 1);  // behind the arithmetic, all four slots being free: see the kernel's header
    }
    __syncthreads();  // the last block of L: nothing is requested behind it
    if (fail_s) return;
    const double *buf = vs[(B - 1) & 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) chunk_fma(buf, c, ar[c], acc);
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  publish(true, brow - acc);
  if (w == 0) TS_STAMP(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's rows of Dinv_B are in LDS (requested first: long there)
  __syncthreads();  // (nothing to order for the granules themselves: this releases the courier's sweep of w_B)
  __syncthreads();  // w_B is in vs[B & 1]
  if (fail_s) return;
  double res = 0.0;
  {
    // the chunks 0 .. sub of Dinv's rows (lower triangular: zeros right of the diagonal), from LDS
    const double *buf = vs[B & 1];
    for (int c = 0; c <= sub; ++c) {
      f64x2 dv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) dv[i] = *reinterpret_cast<const f64x2 *>(dlw + (c * 8 + i) * 128 + 2 * lane);
      chunk_fma(buf, c, dv, res);
    }
  }
  res += __shfl_xor(res, 16);
  res += __shfl_xor(res, 32);
  publish(false, res);
  if (w == 0) TS_STAMP(5);
  // |v|^2 over this workgroup's rows: lanes 0..15 of each wave hold one row each
  double s2 = (fk == 0) ? res * res : 0.0;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
  if (lane == 0) red[w] = s2;
  __syncthreads();
  if (tid == 0) g.sqpart[wg] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the one-right-hand-side form; ws as for launch_trsm_small (its layout is this function's own)
static void launch_trsv_granule(hipStream_t s, const double *L, int64_t ld, const double *Dinv, const double *b,
                                int64_t npad, void *ws, double *dq, unsigned **tmo_dev) {
  const int nb = (int)(npad / PANEL), nwg = (int)(npad / TS_RT);
  char *p = (char *)ws;
  // [tmo (16 B) | Vg | Wg]: one block from the start of the allocation, a multiple of 16 bytes, zeroed per call
  const size_t zbytes = 16 + 2 * (size_t)npad * 16;
  (void)rec_memset_async(p, 0, zbytes, s);
  T1Args g;
  g.L = L;
  g.ld = (long)ld;
  g.Dinv = Dinv;
  g.b = b;
  g.tmo = (unsigned *)p;
  g.Vg = (u32x4 *)(p + 16);
  g.Wg = g.Vg + npad;
  g.sqpart = (double *)(p + ((zbytes + 255) / 256 * 256));
  g.nb = nb;
  g.stamps = g_ts_stamps;
  if (tmo_dev) *tmo_dev = g.tmo;
  const size_t lds = (size_t)(TS_DINV_LDS + 2 * PANEL + 4 + 2) * sizeof(double);
  static bool raised = false;  // 132 KB of dynamic LDS: above the default limit
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&trsv_granule_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    raised = true;
  }
  GOGP_KLAUNCH(trsv_granule_kernel, dim3(nwg), dim3(320), lds, s, g);
  GOGP_KLAUNCH(trsm_small_finish_kernel, dim3(1), dim3(64), 0, s, (const double *)g.sqpart, nwg, 1, 1, 0, dq);
}

// workspace of ONE launch (<= 32 right-hand sides): counters | solution blocks | w blocks | partial sums
size_t trsm_small_workspace_bytes(int64_t npad) {
  const size_t nb = (size_t)(npad / PANEL), nwg = (size_t)(npad / TS_RT);
  return 2 * (size_t)npad * 32 * sizeof(double) + nwg * 32 * sizeof(double) + (2 * nb + 8) * sizeof(unsigned) + 1024;
}

// V = L^-1 Kstar for the right-hand sides j0 .. j0 + cnt - 1 (rows of KsT, cnt <= 32); dq[j] = |V_j|^2 for those j.
// ws: trsm_small_workspace_bytes(npad) bytes of this launch's own.  *tmo_dev: device word that is non-zero afterwards
// if a workgroup gave up waiting (the caller copies it back and reports GOGP_EHIP).
template <class TL>
static void launch_trsm_small_t(hipStream_t s, const TL *L, int64_t ld, const TL *Dinv, const TL *KsT, int64_t ldk,
                                int64_t npad, int j0, int cnt, void *ws, double *dq, unsigned **tmo_dev) {
  const int nb = (int)(npad / PANEL), nwg = (int)(npad / TS_RT);
  char *p = (char *)ws;
  unsigned *flags = (unsigned *)p;  // [cnt | done | tmo ...]: one block at the start of the allocation, zeroed per call
  const size_t fbytes = ((size_t)(2 * nb + 8) * sizeof(unsigned) + 15) / 16 * 16;
  p += (fbytes + 255) / 256 * 256;
  (void)rec_memset_async(flags, 0, fbytes, s);
  if (tmo_dev) *tmo_dev = flags + 2 * nb;
  TsArgs<TL> g;
  g.L = L;
  g.ld = (long)ld;
  g.Dinv = Dinv;
  g.KsT = KsT;
  g.ldk = (long)ldk;
  g.j0 = j0;
  g.Vk = (double *)p;
  p += (size_t)npad * 32 * sizeof(double);
  g.Wk = (double *)p;
  p += (size_t)npad * 32 * sizeof(double);
  g.sqpart = (double *)p;
  g.cnt = flags;
  g.done = flags + nb;
  g.tmo = flags + 2 * nb;
  g.nb = nb;
  const int m_end = j0 + cnt;
#define GOGP_TS(NTV, MCV)                                                                                              \
  do {                                                                                                                 \
    const size_t lds = (size_t)PANEL * 16 * (NTV) * sizeof(double) + 16;                                               \
    if (lds > 64 * 1024 - 1) {                                                                                         \
      static bool raised = false; /* 64 KB + 16 B of dynamic LDS: above the default limit */                           \
      if (!raised) {                                                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&trsm_small_kernel<NTV, MCV, TL>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        raised = true;                                                                                                 \
      }                                                                                                                \
    }                                                                                                                  \
    GOGP_KLAUNCH((trsm_small_kernel<NTV, MCV, TL>), dim3(nwg), dim3(256), lds, s, g);                                        \
    GOGP_KLAUNCH(trsm_small_finish_kernel, dim3(1), dim3(64), 0, s, (const double *)g.sqpart, nwg, 16 * (NTV), m_end, \
                 j0, dq);                                                                                              \
  } while (0)
  if (cnt <= 1) GOGP_TS(1, 1);
  else if (cnt <= 2) GOGP_TS(1, 2);
  else if (cnt <= 4) GOGP_TS(1, 4);
  else if (cnt <= 8) GOGP_TS(1, 8);
  else if (cnt <= 16) GOGP_TS(1, 16);
  else GOGP_TS(2, 32);
#undef GOGP_TS
}

void launch_trsm_small(hipStream_t s, const double *L, int64_t ld, const double *Dinv, const double *KsT, int64_t ldk,
                       int64_t npad, int j0, int cnt, void *ws, double *dq, unsigned **tmo_dev) {
  if (j0 == 0 && cnt == 1) {  // one right-hand side: the data-tagged granule kernel
    launch_trsv_granule(s, L, ld, Dinv, KsT, npad, ws, dq, tmo_dev);
    return;
  }
  launch_trsm_small_t<double>(s, L, ld, Dinv, KsT, ldk, npad, j0, cnt, ws, dq, tmo_dev);
}
// fp32 path: float factor, block inverses and right-hand sides; the counter kernel for every M (one right-hand side too:
// the granule kernel's LDS-DMA images are laid out for doubles)
void launch_trsm_small(hipStream_t s, const float *L, int64_t ld, const float *Dinv, const float *KsT, int64_t ldk,
                       int64_t npad, int j0, int cnt, void *ws, double *dq, unsigned **tmo_dev) {
  launch_trsm_small_t<float>(s, L, ld, Dinv, KsT, ldk, npad, j0, cnt, ws, dq, tmo_dev);
}

}  // namespace gogp
