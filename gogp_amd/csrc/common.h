// common.h -- shared declarations of libgogp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/gogp_hip.h"
#include "graphrec.h"

namespace gogp {

// Blocking constants.  TILE is the GEMM workgroup tile and the diagonal-block
// size; PANEL is the rank of one trailing update (2 diagonal blocks).
constexpr int TILE = 128;
constexpr int PANEL = 256;
constexpr int GEMM_BK = 16;

// Device-side copy of the kernel description + current hyperparameters.
// Lives in device memory; every field is read with wave-uniform (scalar) loads.
struct DevParams {
  int ndim, nterms, ns, nn;
  int kind[GOGP_MAX_TERMS];
  int ard[GOGP_MAX_TERMS];
  double c[GOGP_MAX_TERMS];  // output scale of the term (1 if it has none)
  double w[GOGP_MAX_TERMS];  // PERIODIC: pi / (period_mult * theta_p)
  double inv_len[GOGP_MAX_TERMS][GOGP_MAX_NDIM];  // 1/l_d (all equal unless ard)
  double noise_var;  // value added on the diagonal
  double dnoise;     // d noise_var / d log(std)  (0 for ConstantNoise)
};

// Accumulator slots of the fused gradient reduction (see grad.hip):
//   slot 3*t+0: d/dlog scale of term t, 3*t+1: d/dlog len (non-ARD),
//   3*t+2: d/dlog period; slot 12: trace(W) (noise); 16+d: ARD length d.
constexpr int ACC_TRACE = 3 * GOGP_MAX_TERMS;
constexpr int ACC_ARD0 = 16;
constexpr int NACC = ACC_ARD0 + GOGP_MAX_NDIM;

// Candidate batching (gogp_observe_gradient_candidates): one launch sequence evaluates k hyper-
// parameter candidates on the same data.  Every per-candidate buffer of candidate c lies
// c * stride BYTES after candidate 0's (one arena slot per candidate, api.hip); the launchers
// below that are on the Observe + Gradient path put the candidate index on gridDim.z and the
// kernels shift their per-candidate pointers by blockIdx.z * stride.  Inputs (X, y) are shared.
// The orchestrating thread sets tl_batch for the duration of the call; {1, 0} otherwise.
struct Batch {
  int k = 1;
  long stride = 0;
};
inline thread_local Batch tl_batch;
template <class P>
__device__ __forceinline__ P *cand(P *p, long bstride) {
  return (P *)((const char *)p + (long)blockIdx.z * bstride);
}

struct GemmProfile {
  bool on = false;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  double flops = 0;
  int64_t launches = 0;
  // per launch, in launch order (the events are pool[2i], pool[2i+1]): flops and a tag
  // mode * 1e6 + tiles-across * 1e3... see gogp_profile_read_launches
  std::vector<double> lflops;
  std::vector<int64_t> ltag;
};

// Tile filter of the tile kernel in a sharded (2-D block-cyclic) evaluation (GemmArgs in
// dgemm.hip): the launch covers LOCAL 128-tiles starting at local row / column block
// rblk0 / cblk0 (a distribution block = 2^tpb_shift tiles); this rank sits at (pr, pc) of
// the Pr x Pc process grid.  rule 1: keep the tiles of the GLOBAL lower triangle; rule 2:
// the same, and tiles of global row block beta0 overwrite C (beta = 0) while the others
// accumulate (beta = 1): the rank-k updates of K^-1 = Y Y^T.
// GEMM_LOWER only: new_row0 >= 0 -- tile rows ti >= new_row0 (128-tiles) overwrite C (beta = 0), the
// rows above them accumulate with the launch's beta: the rank-k updates K^-1 (+)= Y_P Y_P^T of the
// fused sweep (api.hip), whose newest block rows have no earlier contribution.
struct GemmGrid {
  int rule = 0, tpb_shift = 0, rblk0 = 0, cblk0 = 0, pr = 0, Pr = 1, pc = 0, Pc = 1, beta0 = -1;
  int new_row0 = -1;
  // GEMM_RECT: B (nt*128 x K) is lower triangular (a block inverse: the panel solves L21 = A21 inv(L11)^T):
  // tile column tj only sums k < (tj + 1) * tile -- the rest of its rows of B is zero
  int ktri = 0;
  // the launch sits on a dependency chain (panel solves, look-ahead updates): its waves raise their issue
  // priority (s_setprio 3), so that on a CU they share with bulk workgroups their MFMAs go first
  int prio = 0;
  // GEMM_RECT / GEMM_LOWER: the rows of A from tile row krag0 (128-tiles) on are upper triangular against the K
  // range -- row block krag0 + i of A is zero for k < i * 128 (a super-panel of Y = L^-T with its triangular diagonal
  // part: api.hip, trtri_superstep) -- so tile row ti >= krag0 only sums k >= (ti - krag0) * 128.  -1: none.
  int krag0 = -1;
  // launches with fewer 128-tiles than this use 64 x 64 tiles (default 384).  Alone on the GPU the 64-tile shape wins
  // up to ~3000 tiles (820 tiles at K = 768: 61 against 51 TFLOP/s); inside an evaluation, beside other streams'
  // launches, it loses -- so only Produce, which runs alone, raises it.
  int small_below = 384;
  // GEMM_LAUUM in two launches (api.hip: option "kinv_split"): tile (ti, tj) sums k >= max(ti * 128, kbeg0) and the
  // tile rows above kbeg0 ACCUMULATE into C (the first launch, K = kbeg0, left their sums over k < kbeg0 there)
  int kbeg0 = 0;
};

// Local <-> global index map of the 2-D block-cyclic layout: distribution blocks of
// nb = 2^nb_shift rows / columns; local row block bi of the rank at grid row pr is global block
// bi * Pr + pr (columns: pc, Pc).
struct BlockMap {
  int nb_shift = 9, pr = 0, Pr = 1, pc = 0, Pc = 1;
  __host__ __device__ long grow(long lrow) const {
    const long nbm = (1L << nb_shift) - 1;
    return ((((lrow >> nb_shift) * Pr + pr)) << nb_shift) + (lrow & nbm);
  }
  __host__ __device__ long gcol(long lcol) const {
    const long nbm = (1L << nb_shift) - 1;
    return ((((lcol >> nb_shift) * Pc + pc)) << nb_shift) + (lcol & nbm);
  }
};

// ---- per-workgroup time stamps (probe build only: make stamp -> tools/exp/lib_stamp.so, -DGOGP_WGSTAMP) -----------------
// where INSIDE a launch of the chain's tile kernel / the diagonal-block kernel the time goes
// when it runs beside the bulk updates.  Every workgroup of those kernels writes s_memrealtime (100 MHz, one clock for the
// whole chip) at entry / operands of the first k-step in LDS / last k-step done / stores drained, plus HW_ID and XCC_ID,
// into a buffer the probe (tools/wg_stamps.py) hands in through these globals; one record per launch says which slice of
// the buffer is whose.  The product library is built without the macro and contains none of this.
#ifdef GOGP_WGSTAMP
struct StampRec {
  long long base, nwg, tag, stream;
};
constexpr int STAMP_SLOTS = 8;
extern unsigned long long *g_stamp_buf;  // device buffer, STAMP_SLOTS x g_stamp_cap entries
extern long long g_stamp_cap, g_stamp_used, g_stamp_nrec;
extern StampRec g_stamp_rec[1 << 16];
// reserve nwg workgroup slots for a launch; nullptr when the probe is not armed or the buffer is full
unsigned long long *stamp_reserve(long long nwg, long long tag, hipStream_t s);
#endif

// ---- launchers implemented in the .hip files ------------------------------
enum GemmMode { GEMM_RECT = 0, GEMM_LOWER = 1, GEMM_LAUUM = 2,
                // GEMM_RECT whose tiles in strictly upper 256x256 blocks (block column > block row,
                // counted from the C origin) are skipped: several adjacent block columns updated
                // "each from its own diagonal block down" in ONE launch
                GEMM_TRAP = 3 };

// C(mt*128 x nt*128) = beta*C + alpha * A * B^T, row-major, K multiple of 16.
// GEMM_LOWER: square tile grid mt x mt, only tiles ti >= tj.
// GEMM_LAUUM: lower tiles; tile (ti,tj) sums k over [ti*128, K).
void launch_dgemm_nt(hipStream_t s, GemmMode mode, int mt, int nt, int64_t K,
                     double alpha, const double *A, int64_t lda, const double *B,
                     int64_t ldb, double beta, double *C, int64_t ldc,
                     GemmProfile *prof, const GemmGrid *grid = nullptr);

// fp32 path (BASELINE config 5): the same launchers overloaded on the MATRIX element type.
// Inputs, vectors (y, z, alpha) and every reduction stay fp64; only the N x N matrices
// (K / L / Y / K^-1, the block inverses, the Produce workspaces) are float.
void launch_gemm_nt(hipStream_t s, GemmMode mode, int mt, int nt, int64_t K, double alpha, const double *A,
                    int64_t lda, const double *B, int64_t ldb, double beta, double *C, int64_t ldc,
                    GemmProfile *prof, const GemmGrid *grid = nullptr);  // = launch_dgemm_nt
void launch_gemm_nt(hipStream_t s, GemmMode mode, int mt, int nt, int64_t K, double alpha, const float *A,
                    int64_t lda, const float *B, int64_t ldb, double beta, float *C, int64_t ldc,
                    GemmProfile *prof, const GemmGrid *grid = nullptr);  // sgemm.hip
void launch_gram_lower_split(hipStream_t s_first, hipStream_t s_rest, const DevParams *p, int ndim,
                             const double *X, int64_t n, int64_t npad, float *K, int64_t ld,
                             int64_t wcols);
void launch_cross(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                  int64_t npad, const double *Z, int64_t m, int64_t mpad, float *KsT, int64_t ld);
void launch_trsv_fwd_step(hipStream_t s, const float *L, int64_t ld, const float *Dinv, int b, int nblk,
                          double *y, double *z);
void launch_trsv_bwd_step(hipStream_t s, const float *L, int64_t ld, const float *Dinv, int b, int nblk,
                          double *zwork, double *alpha);
void launch_lml_scalars(hipStream_t s, const float *L, int64_t ld, const double *z, const double *y,
                        const double *alpha, int64_t n, double *scalars);
void launch_alpha_from_y(hipStream_t s, const float *Y, int64_t ld, const double *z, int64_t npad,
                         double *alpha);
void launch_rownorm_dot(hipStream_t s, const float *V, int64_t ld, const double *vec, int64_t ncols,
                        int64_t m, double *dot, double *sq);
void launch_zero_upper_blocks(hipStream_t s, float *R, int64_t ld, int64_t npad);
void launch_zero_block(hipStream_t s, float *B, int64_t ld, int64_t rows, int64_t cols);
void launch_ydiag(hipStream_t s, const float *Dinv, float *Ydiag, int64_t ld);
void launch_extract_lower(hipStream_t s, const float *L, int64_t ld, int64_t n, double *out);
void launch_pack_lower(hipStream_t s, const double *in, int64_t n, int64_t npad, float *L, int64_t ld);
void launch_convert_block(hipStream_t s, const float *src, int64_t lds_, double *dst, int64_t ldd, int rows,
                          int cols);
void launch_convert_block(hipStream_t s, const double *src, int64_t lds_, float *dst, int64_t ldd, int rows,
                          int cols);
void launch_grad_reduce(hipStream_t s, const DevParams *p, int ndim, int ard_dims, const double *X,
                        const double *alpha, const float *Kinv, int64_t ld, int64_t n, int64_t npad,
                        double *partials, double *out, bool radial1 = false, int mfma_min_dims = 1);

void launch_gram_lower(hipStream_t s, const DevParams *p, int ndim, const double *X,
                       int64_t n, int64_t npad, double *K, int64_t ld);
// Local tiles (mrows x ncols) of a 2-D block-cyclic Gram matrix: tiles of the global lower
// triangle get kernel values (identity padding for rows >= n), distribution blocks strictly
// above the diagonal are zero-filled (the work area R of the triangular inverse).
void launch_gram_local(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                       int64_t mrows, int64_t ncols, BlockMap map, double *K, int64_t ld);
void launch_gram_local(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                       int64_t mrows, int64_t ncols, BlockMap map, float *K, int64_t ld);
void launch_gram_lower_split(hipStream_t s_first, hipStream_t s_rest, const DevParams *p, int ndim,
                             const double *X, int64_t n, int64_t npad, double *K, int64_t ld,
                             int64_t wcols);
// KsT (mpad x npad): KsT[j][i] = k(x_i, z_j); zero for i >= n or j >= m.
void launch_cross(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                  int64_t npad, const double *Z, int64_t m, int64_t mpad, double *KsT,
                  int64_t ld);
// r = y - K v with K recomputed in fp64 on the fly (iterative refinement of alpha on the fp32 path);
// part: nslab * npad doubles of scratch
void launch_residual(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n, int64_t npad,
                     const double *v, const double *y, double *part, int nslab, double *r, bool radial1 = false);
void launch_kmatvec_share(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n, int64_t npad,
                          const double *v, int part_idx, int nparts, double *part, int nslab, double *out,
                          bool radial1 = false);
void launch_prior(hipStream_t s, const DevParams *p, const double *Z, int64_t m,
                  double *prior);

// forward / backward substitution steps with the stored 256x256 block inverses
// (b, nb count 256-blocks)
void launch_trsv_fwd_step(hipStream_t s, const double *L, int64_t ld, const double *Dinv,
                          int b, int nblk, double *y, double *z);
void launch_trsv_bwd_step(hipStream_t s, const double *L, int64_t ld, const double *Dinv,
                          int b, int nblk, double *zwork, double *alpha);

// scalars[0] = sum_i 2 log L_ii, scalars[1] = sum z_i^2, scalars[2] = sum y_i alpha_i
// (i < n; the last only when alpha != nullptr)
void launch_lml_scalars(hipStream_t s, const double *L, int64_t ld, const double *z,
                        const double *y, const double *alpha, int64_t n, double *scalars);
// potrf + dense inverse of one 256x256 diagonal block (diag256.hip); Dinv has
// leading dimension 256.  A (ld): in: lower triangle of the block; Lout (ldl): out: lower
// factor (upper zeroed); Dinv: out: dense inverse of the factor (upper zero).  info: device
// int64, set to (row0+j+1) at the first non-positive pivot with row0+j < nvalid (first
// failure wins).
void launch_diag256(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl,
                    double *Dinv, int64_t row0, int64_t nvalid, long long *info);
// the same with Dinv written into a 256x256 sub-block of a matrix of leading dimension 512
void launch_diag256_ld512(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl,
                          double *Dinv, int64_t row0, int64_t nvalid, long long *info);
// one 128 x 128 half (0 / 1) of the 256-block: factor + its inverse into the block's own quarter of Lout / Dinv (option
// "chain_split": the tile kernel does the products between the halves, api.hip)
void launch_diag128(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl, double *Dinv, int half,
                    int64_t row0, int64_t nvalid, long long *info);
void launch_diag256_inv_only(hipStream_t s, const double *L, int64_t ld, double *Dinv);
// diagsyrk.hip (fp32 path, option "diag_fp64"): D64 block b (256 x 256 doubles, b < nblocks) -= R_b R_b^T with R_b the
// 256 x K float rows at Lrows + b * 256 * ld, summed in fp64; and the strip's start: the float diagonal blocks widened
void launch_diag_syrk_f64(hipStream_t s, const float *Lrows, int64_t ld, int64_t K, double *D64, int nblocks);
void launch_widen_diag_blocks(hipStream_t s, const float *A, int64_t ld, double *D64, int nblocks);
// the same for blocks of bs x bs (256 or 512) whose rows start row_stride floats apart (the diagonal tiles of a shard)
void launch_diag_syrk_f64_tiles(hipStream_t s, const float *Lrows, int64_t ld, int64_t K, double *D64, int nblocks,
                                int64_t row_stride, int bs);
// panel128.hip (option "chain_split" = 2): one 128-column step of the Cholesky chain in one launch -- the diagonal 128-block of
// half 0 / 1 of the 256-block at A is factored and the rows_below (multiple of 64) rows under it are solved against it
void launch_panel128(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl, int half, int64_t rows_below,
                     int64_t row0, int64_t nvalid, long long *info);
// the same with the number of 64-row slabs per workgroup forced (0: by size); the result does not depend on it
void launch_panel128_slabs(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl, int half,
                           int64_t rows_below, int64_t row0, int64_t nvalid, long long *info, int slabs);
// diag256.hip: a whole evaluation of n <= 128 observations in one launch (npad = 256): Gram matrix, factor, block inverse, z,
// alpha and (want_kinv) K^-1, left where the general path leaves them (A: K^-1 lower, L, Dinv: leading dimension 256)
void launch_tiny_eval(hipStream_t s, const DevParams *P, const double *X, const double *y, int64_t n, double *A, double *L,
                      double *Dinv, double *z, double *alpha, long long *info, bool want_kinv);
// dense inverses of nblk consecutive 256 x 256 diagonal blocks of a finished factor (block b at L + b * 256 * (ld + 1))
void launch_dinv256_blocks(hipStream_t s, const double *L, int64_t ld, double *Dinv, int nblk);
void launch_diag256_inv_only_ld512(hipStream_t s, const double *L, int64_t ld, double *Dinv);  // Dinv: ld 512
void launch_convert_block(hipStream_t s, const double *src, int64_t lds_, double *dst, int64_t ldd, int rows,
                          int cols);
void launch_pack_lower(hipStream_t s, const double *in, int64_t n, int64_t npad, double *L,
                       int64_t ld);
void launch_sigma(hipStream_t s, const double *prior, const double *q, int64_t m,
                  double *sigma);
int grad_reduce_blocks(int64_t npad);

// fused gradient reduction over lower tiles of Kinv; out: NACC doubles
void launch_grad_reduce(hipStream_t s, const DevParams *p, int ndim, int ard_dims,
                        const double *X, const double *alpha, const double *Kinv, int64_t ld, int64_t n,
                        int64_t npad, double *partials, double *out, bool radial1 = false, int mfma_min_dims = 1);
// the same over the LOCAL tiles (mrows x ncols, leading dimension ld) of a 2-D block-cyclic
// K^-1: tiles of the global lower triangle only; `partials` needs grad_reduce_blocks_local
int grad_reduce_blocks_local(int64_t mrows, int64_t ncols);
void launch_grad_reduce_local(hipStream_t s, const DevParams *p, int ndim, int ard_dims,
                              const double *X, const double *alpha, const double *Kinv, int64_t ld,
                              int64_t n, int64_t mrows, int64_t ncols, BlockMap map, double *partials,
                              double *out, bool radial1 = false, int mfma_min_dims = 1);
void launch_grad_reduce_local(hipStream_t s, const DevParams *p, int ndim, int ard_dims,
                              const double *X, const double *alpha, const float *Kinv, int64_t ld,
                              int64_t n, int64_t mrows, int64_t ncols, BlockMap map, double *partials,
                              double *out, bool radial1 = false, int mfma_min_dims = 1);

// grad_mfma.hip: the reduction pass for ONE radial term with ARD length scales, distances and per-dimension sums
// on the matrix cores; ntc == 0: lower triangle of an unsharded K^-1 (nt x nt tiles of 64, candidate batching
// honoured), ntc > 0: the local nt x ntc tiles of a 2-D block-cyclic one.  Writes `blocks` x NACC partials.
void launch_grad_ard_mfma(hipStream_t s, const DevParams *p, int ndim, const double *X, const double *alpha,
                          const double *Kinv, int64_t ld, int64_t n, int nt, int ntc, int ntiles, int blocks,
                          BlockMap map, double *partials);
void launch_grad_ard_mfma(hipStream_t s, const DevParams *p, int ndim, const double *X, const double *alpha,
                          const float *Kinv, int64_t ld, int64_t n, int nt, int ntc, int ntiles, int blocks,
                          BlockMap map, double *partials);
// (ARD kernels with at least `mfma_min_dims` dimensions take it -- option "ard_mfma_min_dims", default 1: measured at
// N = 16384 it beats the scalar-row kernel of grad.hip at every D: 8: 1.16 -> 0.76 ms, 16: 1.91 -> 0.76, 32: 3.63 -> 1.01,
// 64: 10.85 -> 2.50; 65 = never)

// gradient w.r.t. the inputs: mirrors K^-1 to the upper triangle, then
// gx[i][d] = sum_j (alpha_i alpha_j - Kinv_ij) dk(x_i,x_j)/dx_{i,d}
void launch_xgrad(hipStream_t s, const DevParams *p, int ndim, const double *X,
                  const double *alpha, double *Kinv, int64_t ld, int64_t n, int64_t npad,
                  double *gx);

// dot_j = sum_i V[j][i] vec_i ; sq_j = sum_i V[j][i]^2  (either output may be null)
void launch_rownorm_dot(hipStream_t s, const double *V, int64_t ld, const double *vec,
                        int64_t ncols, int64_t m, double *dot, double *sq);

void launch_zero_upper_blocks(hipStream_t s, double *R, int64_t ld, int64_t npad);
void launch_alpha_from_y(hipStream_t s, const double *Y, int64_t ld, const double *z,
                         int64_t npad, double *alpha);
void launch_zero_block(hipStream_t s, double *B, int64_t ld, int64_t rows, int64_t cols);
void launch_ydiag(hipStream_t s, const double *Dinv, double *Ydiag, int64_t ld);
// T^-1 of a super-panel's diagonal block (api.hip: assemble_tinv): diagonal / zero blocks, and nprod (<= 6) products
// C_b (256 x 256) = alpha A_b (256 x K_b) B_b (K_b x 256), row-major, in one launch (solve.hip)
void launch_tinv_init(hipStream_t s, const double *Dinv, double *X, double *XT, int nsub, int64_t tld);
void launch_tinv_init(hipStream_t s, const float *Dinv, float *X, float *XT, int nsub, int64_t tld);
void launch_blockmm(hipStream_t s, int nprod, const double *const *A, const int64_t *lda, const double *const *B,
                    const int64_t *ldb, double *const *C, const int64_t *ldc, const int *K, double alpha, int side = 256);
void launch_blockmm(hipStream_t s, int nprod, const float *const *A, const int64_t *lda, const float *const *B,
                    const int64_t *ldb, float *const *C, const int64_t *ldc, const int *K, double alpha);
// fp32 path: out[0] = sum_{i<n} alpha_i^2 - |Y|_F^2 (= tr(alpha alpha^T - K^-1)) in fp64 from Y = L^-T; part: n doubles
void launch_trace_from_y(hipStream_t s, const float *Y, int64_t ld, int64_t n, int64_t npad, const double *alpha,
                         double *part, double *out);
// trsm_small.hip: V = L^-1 Kstar for the right-hand sides j0 .. j0 + cnt - 1 (rows of KsT, cnt <= 32) in ONE persistent
// launch that reads the factor once; dq[j] = |V_j|^2.  ws: trsm_small_workspace_bytes(npad) bytes of this launch's own;
// *tmo_dev: device word that is non-zero afterwards if a workgroup gave up waiting (the caller copies it back and
// reports GOGP_EHIP)
size_t trsm_small_workspace_bytes(int64_t npad);
void launch_trsm_small(hipStream_t s, const double *L, int64_t ld, const double *Dinv, const double *KsT, int64_t ldk,
                       int64_t npad, int j0, int cnt, void *ws, double *dq, unsigned **tmo_dev);
void launch_trsm_small(hipStream_t s, const float *L, int64_t ld, const float *Dinv, const float *KsT, int64_t ldk,
                       int64_t npad, int j0, int cnt, void *ws, double *dq, unsigned **tmo_dev);  // fp32 path
void launch_fill(hipStream_t s, double *p, int64_t count, double v);
void launch_axpy(hipStream_t s, double *a, const double *b, int64_t count);  // a += b
void launch_dot(hipStream_t s, const double *a, const double *b, int64_t n, double *out);  // out[0] = a.b
// helpers of the sharded evaluation (solve.hip)
void launch_transpose_sq(hipStream_t s, const double *src, int64_t lds_, double *dst, int64_t ldd,
                         int n);
void launch_transpose_sq(hipStream_t s, const float *src, int64_t lds_, float *dst, int64_t ldd, int n);
void launch_pack_blocks(hipStream_t s, double *dst, const double *src, int nblk, int64_t blk,
                        int first, int stride);
int64_t chunk_tdot_scratch(int64_t max_rows, int nb);  // doubles of `part` scratch
void launch_chunk_tdot(hipStream_t s, const double *chunk, int64_t rows, int nb, const double *v,
                       double *part, double *out);
void launch_chunk_tdot(hipStream_t s, const float *chunk, int64_t rows, int nb, const double *v,
                       double *part, double *out);
void launch_chunk_alpha(hipStream_t s, const double *Ych, int mloc, int nloc, int nb, BlockMap map,
                        const double *z, double *out);
void launch_chunk_alpha(hipStream_t s, const float *Ych, int mloc, int nloc, int nb, BlockMap map,
                        const double *z, double *out);
// out[0] = |Y_local|_F^2 over this rank's chunks of Y (rows < n), fp64; part: mloc * nb doubles
void launch_chunk_sumsq(hipStream_t s, const double *Ych, int mloc, int nloc, int nb, BlockMap map, int64_t n, double *part,
                        double *out);
void launch_chunk_sumsq(hipStream_t s, const float *Ych, int mloc, int nloc, int nb, BlockMap map, int64_t n, double *part,
                        double *out);
void launch_logdet_block(hipStream_t s, const double *L, int64_t ld, int64_t row0, int64_t n, int nb,
                         double *acc);
void launch_sumsq_info(hipStream_t s, const double *z, int64_t n, const long long *info, double *out);
void launch_info_to_double(hipStream_t s, const long long *info, double *out);
void launch_extract_lower(hipStream_t s, const double *L, int64_t ld, int64_t n,
                          double *out);

}  // namespace gogp
