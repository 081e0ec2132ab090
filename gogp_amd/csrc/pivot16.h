// pivot16.h -- the 16 x 16 base case of the in-LDS Cholesky factorisations (panel128.hip, diag256.hip): one wave factors
// the pivot block and forward-substitutes one panel row per lane in the same pass, with DPP broadcasts inside rows of 16
// lanes and the rank-1 updates software-pipelined into the latencies of the pivot chain.
//
// Reference counterpart: the unblocked Dpotf2 / Dtrsm at the bottom of gonum's blocked Dpotrf (mat.Cholesky.Factorize,
// gp/gp.go:228).
#pragma once
#include <hip/hip_runtime.h>

namespace gogp {

// x of lane C of the caller's own row of 16 lanes (DPP row_newbcast: one v_mov_b64_dpp, no trip through SGPRs; all lanes
// are written, so there is no previous value to keep: bound_ctrl)
template <int C>
__device__ __forceinline__ double bcast16(double x) {
  return __builtin_amdgcn_update_dpp(x, x, 0x150 + C, 0xf, 0xf, true);
}

// rank-1 update of columns C0 .. C1 - 1 by a pivot column: the pivot block's rows (ad) and the panel rows (ap) share the
// broadcast of the column's value in row C
template <int C0, int C1>
struct Fill {
  static __device__ __forceinline__ void run(double (&ad)[16], double (&ap)[16], double ld, double lp) {
    if constexpr (C0 < C1) {
      const double b = bcast16<C0>(ld);
      ad[C0] = fma(-ld, b, ad[C0]);
      ap[C0] = fma(-lp, b, ap[C0]);
      Fill<C0 + 1, C1>::run(ad, ap, ld, lp);
    }
  }
};

// Pivot column J of the 16 x 16 block.  The dependency chain from one pivot to the next is
//   d -> v_rsq_f64 -> two Newton steps (three dependent operations each, the last one folded into the scaling of the
//   column: l = g + g e with g = a rs, e = 1/2 - (d/2) rs^2) -> broadcast of l_{J+1,J} -> next d = a_{J+1,J+1} - l^2
// -- nine dependent instructions of a wave that issues in order.  Everything else, the rank-1 update of columns
// J + 2 .. 15 (of the pivot block's rows and of the panel rows alike), is NOT on it: the update by column J - 1 (lq, pq)
// is dealt in eight chunks into the latencies of column J's chain, and scheduling barriers keep hipcc from undoing the
// interleave (left alone it hoists every broadcast to the top: 267 VGPR spills, or sinks every update to the column that
// needs it: a second dependent chain).
// RINV: lane r of every row of 16 lanes also keeps 1/L_rr of ITS pivot row in `mine` (the 16 x 16 block inverses of
// diag256.hip start from it).  SAVE: the column's two scaling factors -- rs after the first Newton step and the last
// correction e -- go to sv[svs J], sv[svs J + 1]: with them SolveColumn below repeats the panel rows' arithmetic on OTHER
// rows, operation for operation (panel128.hip: further slabs of rows per workgroup).  Branch-free: ONE lane is handed the
// real table with svs = 2, every other lane a two-double scratch of its own with svs = 0 (a branch per column would cut
// the pass into sixteen basic blocks and undo its schedule: 636 VGPR spills).
template <int J, bool RINV, bool SAVE = false>
struct PivotColumn {
  static __device__ __forceinline__ void run(double (&ad)[16], double (&ap)[16], double d, double lq, double pq, int &bad,
                                             double &mine, int r, double *sv = nullptr, int svs = 0) {
    constexpr int n = J > 0 ? 15 - J : 0, c0 = J + 1;  // pending: columns J + 1 .. 15 of the update by column J - 1
#define GOGP_CHUNK(k)                                                              \
  Fill<c0 + ((k) * n + 7) / 8, c0 + (((k) + 1) * n + 7) / 8>::run(ad, ap, lq, pq); \
  __builtin_amdgcn_sched_barrier(0)
    // off the chain: a non-positive (or NaN) pivot is only recorded; its rsq is NaN / inf and poisons the rest of the
    // factor, which the caller discards (GOGP_ENOTPD)
    bad = (!(d > 0.0) && bad == 16) ? J : bad;
    const double nhd = -0.5 * d;
    double rs = __builtin_amdgcn_rsq(d);
    GOGP_CHUNK(0);  // (never empty while columns are pending: column J + 1 first, the next pivot below reads it)
    double t = nhd * rs;
    GOGP_CHUNK(1);
    double e = fma(t, rs, 0.5);
    GOGP_CHUNK(2);
    rs = fma(rs, e, rs);
    GOGP_CHUNK(3);
    t = nhd * rs;
    const double gd = ad[J] * rs, gp = ap[J] * rs;
    GOGP_CHUNK(4);
    e = fma(t, rs, 0.5);
    GOGP_CHUNK(5);
    const double ld = fma(gd, e, gd), lp = fma(gp, e, gp);  // lane J of the pivot rows: d / sqrt(d) = sqrt(d)
    ad[J] = ld;
    ap[J] = lp;
    if (RINV) mine = (r == J) ? fma(rs, e, rs) : mine;
    if (SAVE) {
      sv[svs * J] = rs;
      sv[svs * J + 1] = e;
    }
    GOGP_CHUNK(6);
    if constexpr (J < 15) {
      const double b = bcast16<J + 1>(ld), bo = bcast16<J + 1>(ad[J + 1]);
      GOGP_CHUNK(7);
      const double dn = fma(-b, b, bo);  // the next pivot
      ad[J + 1] = fma(-ld, b, ad[J + 1]);
      ap[J + 1] = fma(-lp, b, ap[J + 1]);
      __builtin_amdgcn_sched_barrier(0);
      PivotColumn<J + 1, RINV, SAVE>::run(ad, ap, dn, ld, lp, bad, mine, r, sv, svs);
    }
#undef GOGP_CHUNK
  }
};

// The panel rows' half of PivotColumn on its own: rows ap against a pivot block that is ALREADY factored (ad: lane r of
// every row of 16 lanes holds row r of the factor block) with the saved scalings sv -- the same multiplications and FMAs on
// ap, in the same order per accumulator, as the pass that factored the block did on its own panel rows: the same bits.
template <int J, int C>
struct SolveFill {
  static __device__ __forceinline__ void run(const double (&ad)[16], double (&ap)[16], double lp) {
    if constexpr (C < 16) {
      ap[C] = fma(-lp, bcast16<C>(ad[J]), ap[C]);
      SolveFill<J, C + 1>::run(ad, ap, lp);
    }
  }
};
template <int J>
struct SolveColumn {
  static __device__ __forceinline__ void run(const double (&ad)[16], double (&ap)[16], const double *sv) {
    const double rs = sv[2 * J], e = sv[2 * J + 1];
    const double gp = ap[J] * rs;
    const double lp = fma(gp, e, gp);
    ap[J] = lp;
    if constexpr (J < 15) {
      SolveFill<J, J + 1>::run(ad, ap, lp);
      __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise hoists all 120 broadcasts to the top: 636 VGPR spills)
      SolveColumn<J + 1>::run(ad, ap, sv);
    }
  }
};

}  // namespace gogp
