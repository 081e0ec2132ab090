// common.h -- shared declarations of libgogp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/gogp_hip.h"

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

struct GemmProfile {
  bool on = false;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  double flops = 0;
  int64_t launches = 0;
};

// Ownership filter for the tile kernel in a sharded evaluation (see GemmArgs):
// n ranks, this rank r, tiles_per_sp 128-wide tile columns per super-panel, col0 =
// global index of the launch's first 128-wide tile column.
struct GemmOwn {
  int n, r, tiles_per_sp, col0;
};

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
                     GemmProfile *prof, const GemmOwn *own = nullptr);

void launch_gram_lower(hipStream_t s, const DevParams *p, int ndim, const double *X,
                       int64_t n, int64_t npad, double *K, int64_t ld);
void launch_gram_lower_split(hipStream_t s_first, hipStream_t s_rest, const DevParams *p, int ndim,
                             const double *X, int64_t n, int64_t npad, double *K, int64_t ld,
                             int64_t wcols);
// KsT (mpad x npad): KsT[j][i] = k(x_i, z_j); zero for i >= n or j >= m.
void launch_cross(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                  int64_t npad, const double *Z, int64_t m, int64_t mpad, double *KsT,
                  int64_t ld);
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
void launch_diag256_inv_only(hipStream_t s, const double *L, int64_t ld, double *Dinv);
void launch_pack_lower(hipStream_t s, const double *in, int64_t n, int64_t npad, double *L,
                       int64_t ld);
void launch_sigma(hipStream_t s, const double *prior, const double *q, int64_t m,
                  double *sigma);
int grad_reduce_blocks(int64_t npad);

// fused gradient reduction over lower tiles of Kinv; out: NACC doubles
void launch_grad_reduce(hipStream_t s, const DevParams *p, int ndim, int ard_dims,
                        const double *X, const double *alpha, const double *Kinv, int64_t ld, int64_t n,
                        int64_t npad, double *partials, double *out, int own_n = 0, int own_r = 0);

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
void launch_fill(hipStream_t s, double *p, int64_t count, double v);
void launch_extract_lower(hipStream_t s, const double *L, int64_t ld, int64_t n,
                          double *out);

}  // namespace gogp
