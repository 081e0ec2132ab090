// solve.hip -- O(N^2) bandwidth-bound pieces around the factor:
//   blocked forward/backward substitution for alpha = K^-1 y
//     (gonum Cholesky.SolveVecTo, call site gp/gp.go:232-236)
//   log-determinant and y^T K^-1 y for LML (gp/gp.go:244-253)
//   row norms / dots for Produce (gp/gp.go:335,341-342,356)
//   small utilities (identity/zero fill for the triangular inverse, factor export)
// All reductions are fixed-order (no atomics): results are bitwise reproducible.
#include "common.h"

namespace gogp {

// 8 consecutive matrix elements as doubles (16-B loads for both element types)
__device__ __forceinline__ void load8(const double *p, double *o) {
  const double2 *q = reinterpret_cast<const double2 *>(p);
  const double2 a = q[0], b = q[1], c = q[2], d = q[3];
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = c.x; o[5] = c.y; o[6] = d.x; o[7] = d.y;
}
__device__ __forceinline__ void load8(const float *p, double *o) {
  const float4 *q = reinterpret_cast<const float4 *>(p);
  const float4 a = q[0], b = q[1];
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void load2(const double *p, double &a, double &b) {
  const double2 v = *reinterpret_cast<const double2 *>(p);
  a = v.x;
  b = v.y;
}
__device__ __forceinline__ void load2(const float *p, double &a, double &b) {
  const float2 v = *reinterpret_cast<const float2 *>(p);
  a = v.x;
  b = v.y;
}
__device__ __forceinline__ void store2zero(double *p) {
  double2 v;
  v.x = 0.0;
  v.y = 0.0;
  *reinterpret_cast<double2 *>(p) = v;
}
__device__ __forceinline__ void store2zero(float *p) {
  float2 v;
  v.x = 0.0f;
  v.y = 0.0f;
  *reinterpret_cast<float2 *>(p) = v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// res[r] = sum_c M[r][c] * v[c] for 128 rows x 256 columns (M row-major, ldm),
// 256-thread workgroup, v[256] and res[128] in LDS.  32 lanes share a row
// (8 consecutive columns each: 2 KB contiguous per row), 2 rows per
// wave-iteration, 5 xor-shuffles per pair of rows.
template <class T>
__device__ __forceinline__ void matvec_128x256(const T *__restrict__ M, long ldm,
                                               const double *v, double *res) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rsub = lane >> 5, cg = lane & 31;
  double vv[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) vv[k] = v[cg * 8 + k];
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int r = wid * 32 + it * 2 + rsub;
    double mm[8];
    load8(M + (long)r * ldm + cg * 8, mm);
    double q = mm[0] * vv[0] + mm[1] * vv[1] + mm[2] * vv[2] + mm[3] * vv[3] + mm[4] * vv[4] +
               mm[5] * vv[5] + mm[6] * vv[6] + mm[7] * vv[7];
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 8);
    q += __shfl_xor(q, 4);
    q += __shfl_xor(q, 2);
    q += __shfl_xor(q, 1);
    if (cg == 0) res[r] = q;
  }
}

// res[c] = sum_r M[r][c] * v[r] for 256 rows x 128 columns, 256 threads
template <class T>
__device__ __forceinline__ void matvec_t_256x128(const T *__restrict__ M, long ldm,
                                                 const double *v, double *res, double *scratch) {
  const int c = threadIdx.x & 127, half = threadIdx.x >> 7;
  double p0 = 0.0, p1 = 0.0;
#pragma unroll 8  // fully unrolled, the 128 loads in flight cost 272 registers (16 of them AGPR spill space)
  for (int r = half * 128; r < half * 128 + 128; r += 2) {
    p0 += (double)M[(long)r * ldm + c] * v[r];
    p1 += (double)M[(long)(r + 1) * ldm + c] * v[r + 1];
  }
  const double p = p0 + p1;
  if (half == 1) scratch[c] = p;
  __syncthreads();
  if (half == 0) res[c] = p + scratch[c];
  __syncthreads();
}

// Forward step b (256-blocks): z_b = Dinv_b w_b ; w_i -= L[i,b] z_b  (i > b).
// Workgroup g handles 128 rows: g = 0,1 the halves of z_b, g >= 2 rows
// (b+1)*256 + (g-2)*128 ...  Every workgroup recomputes z_b (Dinv_b is L2-resident).
template <class T>
__global__ __launch_bounds__(256) void trsv_fwd_kernel(const T *__restrict__ L, long ld,
                                                       const T *__restrict__ Dinv, int b,
                                                       double *__restrict__ w,
                                                       double *__restrict__ z, long bstride) {
  __shared__ double vb[256], zb[256], upd[128];
  L = cand(L, bstride);  // candidate batching (common.h: Batch)
  Dinv = cand(Dinv, bstride);
  w = cand(w, bstride);
  z = cand(z, bstride);
  const int tid = threadIdx.x;
  const int g = blockIdx.x;
  vb[tid] = w[(long)b * 256 + tid];
  __syncthreads();
  const T *Db = Dinv + (long)b * 256 * 256;
  if (g < 2) {
    matvec_128x256(Db + (long)g * 128 * 256, 256, vb, zb);
    __syncthreads();
    if (tid < 128) z[(long)b * 256 + g * 128 + tid] = zb[tid];
    return;
  }
  matvec_128x256(Db, 256, vb, zb);
  matvec_128x256(Db + 128 * 256, 256, vb, zb + 128);
  __syncthreads();
  const long r0 = (long)(b + 1) * 256 + (long)(g - 2) * 128;
  matvec_128x256(L + r0 * ld + (long)b * 256, ld, zb, upd);
  __syncthreads();
  if (tid < 128) w[r0 + tid] -= upd[tid];
}

// Backward step b: alpha_b = Dinv_b^T w_b ; w_i -= L[b,i]^T alpha_b  (i < b).
// Workgroup g handles 128 columns: g = 0,1 the halves of alpha_b, g >= 2
// columns (g-2)*128 ... of the block row b of L.
template <class T>
__global__ __launch_bounds__(256) void trsv_bwd_kernel(const T *__restrict__ L, long ld,
                                                       const T *__restrict__ Dinv, int b,
                                                       double *__restrict__ w,
                                                       double *__restrict__ alpha) {
  __shared__ double vb[256], ab[256], upd[128], scratch[128];
  const int tid = threadIdx.x;
  const int g = blockIdx.x;
  vb[tid] = w[(long)b * 256 + tid];
  __syncthreads();
  const T *Db = Dinv + (long)b * 256 * 256;
  if (g < 2) {
    matvec_t_256x128(Db + g * 128, 256, vb, ab, scratch);
    if (tid < 128) alpha[(long)b * 256 + g * 128 + tid] = ab[tid];
    return;
  }
  matvec_t_256x128(Db, 256, vb, ab, scratch);
  matvec_t_256x128(Db + 128, 256, vb, ab + 128, scratch);
  const long c0 = (long)(g - 2) * 128;
  matvec_t_256x128(L + (long)b * 256 * ld + c0, ld, ab, upd, scratch);
  if (tid < 128) w[c0 + tid] -= upd[tid];
}

// nb = number of 256-blocks
void launch_trsv_fwd_step(hipStream_t s, const double *L, int64_t ld, const double *Dinv,
                          int b, int nb, double *w, double *z) {
  GOGP_KLAUNCH(trsv_fwd_kernel<double>, dim3(2 + 2 * (nb - b - 1), 1, (unsigned)tl_batch.k), dim3(256), 0,
                     s, L, (long)ld, Dinv, b, w, z, tl_batch.stride);
}
void launch_trsv_fwd_step(hipStream_t s, const float *L, int64_t ld, const float *Dinv,
                          int b, int nb, double *w, double *z) {
  GOGP_KLAUNCH(trsv_fwd_kernel<float>, dim3(2 + 2 * (nb - b - 1)), dim3(256), 0, s, L, (long)ld,
                     Dinv, b, w, z, 0L);
}

void launch_trsv_bwd_step(hipStream_t s, const double *L, int64_t ld, const double *Dinv,
                          int b, int nb, double *w, double *alpha) {
  (void)nb;
  GOGP_KLAUNCH(trsv_bwd_kernel<double>, dim3(2 + 2 * b), dim3(256), 0, s, L, (long)ld, Dinv, b, w,
                     alpha);
}
void launch_trsv_bwd_step(hipStream_t s, const float *L, int64_t ld, const float *Dinv,
                          int b, int nb, double *w, double *alpha) {
  (void)nb;
  GOGP_KLAUNCH(trsv_bwd_kernel<float>, dim3(2 + 2 * b), dim3(256), 0, s, L, (long)ld, Dinv, b, w,
                     alpha);
}

// alpha = Y z with Y = L^-T upper triangular (row-major): alpha_i = sum_{q >= i0(i)} Y[i][q] z[q],
// where i0 = first column of row i's 256-block (the block-lower part of Y is never
// written).  One workgroup per 4 rows (one wave per row), 16-B loads.
template <class T>
__global__ __launch_bounds__(256) void alpha_from_y_kernel(const T *__restrict__ Y, long ld,
                                                           const double *__restrict__ z, long npad,
                                                           double *__restrict__ alpha, long bstride) {
  Y = cand(Y, bstride);  // candidate batching (common.h: Batch)
  z = cand(z, bstride);
  alpha = cand(alpha, bstride);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 4 + wid;
  if (i >= npad) return;
  const long q0 = (i / PANEL) * PANEL;
  const T *row = Y + i * ld;
  double s0 = 0.0, s1 = 0.0;
  for (long q = q0 + lane * 2; q < npad; q += 128) {
    double ya, yb;
    load2(row + q, ya, yb);
    const double2 zz = *reinterpret_cast<const double2 *>(z + q);
    s0 += ya * zz.x;
    s1 += yb * zz.y;
  }
  double s = s0 + s1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) alpha[i] = s;
}

void launch_alpha_from_y(hipStream_t s, const double *Y, int64_t ld, const double *z,
                         int64_t npad, double *alpha) {
  if (npad <= 0) return;
  GOGP_KLAUNCH(alpha_from_y_kernel<double>, dim3((unsigned)((npad + 3) / 4), 1, (unsigned)tl_batch.k),
                     dim3(256), 0, s, Y, (long)ld, z, (long)npad, alpha, tl_batch.stride);
}
void launch_alpha_from_y(hipStream_t s, const float *Y, int64_t ld, const double *z,
                         int64_t npad, double *alpha) {
  if (npad <= 0) return;
  GOGP_KLAUNCH(alpha_from_y_kernel<float>, dim3((unsigned)((npad + 3) / 4)), dim3(256), 0, s, Y,
                     (long)ld, z, (long)npad, alpha, 0L);
}

// scalars[0] = sum_{i<n} 2 log L_ii ; scalars[1] = sum_{i<n} z_i^2 ;
// scalars[2] = sum_{i<n} y_i alpha_i (only if alpha != nullptr); scalars[3], [4] = min, max L_ii
template <class T>
__global__ __launch_bounds__(1024) void lml_scalars_kernel(const T *__restrict__ L, long ld,
                                                           const double *__restrict__ z,
                                                           const double *__restrict__ y,
                                                           const double *__restrict__ alpha,
                                                           long n, double *__restrict__ scalars,
                                                           long bstride) {
  L = cand(L, bstride);  // candidate batching (common.h: Batch); y is shared
  z = cand(z, bstride);
  if (alpha) alpha = cand(alpha, bstride);
  scalars = cand(scalars, bstride);
  __shared__ double red[5][16];
  double a = 0.0, b = 0.0, c = 0.0, dmin = INFINITY, dmax = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) {
    const double lii = (double)L[i * ld + i];
    a += 2.0 * log(lii);
    dmin = fmin(dmin, lii);
    dmax = fmax(dmax, lii);
    const double zi = z[i];
    b += zi * zi;
    if (alpha) c += y[i] * alpha[i];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dmin = fmin(dmin, __shfl_xor(dmin, o));
    dmax = fmax(dmax, __shfl_xor(dmax, o));
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wid] = a;
    red[1][wid] = b;
    red[2][wid] = c;
    red[3][wid] = dmin;
    red[4][wid] = dmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0, sc = 0, mn = INFINITY, mx = 0.0;
    for (int w = 0; w < 16; ++w) {
      sa += red[0][w];
      sb += red[1][w];
      sc += red[2][w];
      mn = fmin(mn, red[3][w]);
      mx = fmax(mx, red[4][w]);
    }
    scalars[0] = sa;
    scalars[1] = sb;
    scalars[2] = sc;
    scalars[3] = mn;  // min / max L_ii: (max/min)^2 bounds cond_2(K) from below
    scalars[4] = mx;
  }
}

void launch_lml_scalars(hipStream_t s, const double *L, int64_t ld, const double *z,
                        const double *y, const double *alpha, int64_t n, double *scalars) {
  GOGP_KLAUNCH(lml_scalars_kernel<double>, dim3(1, 1, (unsigned)tl_batch.k), dim3(1024), 0, s, L,
                     (long)ld, z, y, alpha, (long)n, scalars, tl_batch.stride);
}
void launch_lml_scalars(hipStream_t s, const float *L, int64_t ld, const double *z,
                        const double *y, const double *alpha, int64_t n, double *scalars) {
  GOGP_KLAUNCH(lml_scalars_kernel<float>, dim3(1), dim3(1024), 0, s, L, (long)ld, z, y, alpha,
                     (long)n, scalars, 0L);
}

// one workgroup per row j < m: dot_j = sum_i V[j][i] vec[i], sq_j = sum_i V[j][i]^2
template <class T>
__global__ __launch_bounds__(256) void rownorm_dot_kernel(const T *__restrict__ V, long ld,
                                                          const double *__restrict__ vec,
                                                          long ncols, double *__restrict__ dot,
                                                          double *__restrict__ sq) {
  __shared__ double red[2][4];
  const long j = blockIdx.x;
  const T *row = V + j * ld;
  double a = 0.0, b = 0.0;
  for (long i = threadIdx.x; i < ncols; i += 256) {
    const double v = (double)row[i];
    if (vec) a += v * vec[i];
    b += v * v;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][wid] = a;
    red[1][wid] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (dot) dot[j] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    if (sq) sq[j] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

void launch_rownorm_dot(hipStream_t s, const double *V, int64_t ld, const double *vec,
                        int64_t ncols, int64_t m, double *dot, double *sq) {
  if (m <= 0) return;
  GOGP_KLAUNCH(rownorm_dot_kernel<double>, dim3((unsigned)m), dim3(256), 0, s, V, (long)ld, vec,
                     (long)ncols, dot, sq);
}
void launch_rownorm_dot(hipStream_t s, const float *V, int64_t ld, const double *vec,
                        int64_t ncols, int64_t m, double *dot, double *sq) {
  if (m <= 0) return;
  GOGP_KLAUNCH(rownorm_dot_kernel<float>, dim3((unsigned)m), dim3(256), 0, s, V, (long)ld, vec,
                     (long)ncols, dot, sq);
}

// fp32 path, gradient: out[0] = sum_{i < n} (alpha_i^2 - sum_{q >= q0(i)} Y[i][q]^2) = |alpha|^2 - tr(K^-1) with
// tr(K^-1) = |Y|_F^2 summed in fp64 from Y = L^-T itself (q0: first column of row i's 256-block; what lies left of it
// is never written) -- instead of the diagonal of the fp32 product Y Y^T, whose entries carry the fp32 accumulation
// of up to N squares.  One workgroup per row writes part[i]; a single workgroup adds them in a fixed order.
template <class T>
__global__ __launch_bounds__(256) void trace_rows_kernel(const T *__restrict__ Y, long ld, long ncols,
                                                         const double *__restrict__ alpha, double *__restrict__ part) {
  __shared__ double red[4];
  const long i = blockIdx.x;
  const long q0 = (i / PANEL) * PANEL;
  const T *row = Y + i * ld;
  double b = 0.0;
  for (long q = q0 + threadIdx.x; q < ncols; q += 256) {
    const double v = (double)row[q];
    b += v * v;
  }
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) part[i] = alpha[i] * alpha[i] - ((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ __launch_bounds__(1024) void sum_kernel(const double *__restrict__ v, long n, double *__restrict__ out) {
  __shared__ double red[16];
  double a = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) a += v[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = t;
  }
}
void launch_trace_from_y(hipStream_t s, const float *Y, int64_t ld, int64_t n, int64_t npad, const double *alpha,
                         double *part, double *out) {
  if (n <= 0) return;
  GOGP_KLAUNCH(trace_rows_kernel<float>, dim3((unsigned)n), dim3(256), 0, s, Y, (long)ld, (long)npad, alpha, part);
  GOGP_KLAUNCH(sum_kernel, dim3(1), dim3(1024), 0, s, (const double *)part, (long)n, out);
}

// Zero the STRICTLY upper 256-block triangle of a matrix: row r, columns
// >= (r/256 + 1)*256.  This is where the right-hand side R of the triangular
// inverse lives; the lower triangle + diagonal blocks (the Cholesky work area)
// are not touched, so the two phases can share the buffer concurrently.
template <class T>
__global__ __launch_bounds__(256) void zero_upper_kernel(T *__restrict__ R, long ld,
                                                         long npad, long bstride) {
  R = cand(R, bstride);  // candidate batching (common.h: Batch)
  const long row = blockIdx.y;
  const long cstart = (row / PANEL + 1) * PANEL;
  const long c = cstart + ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (c >= npad) return;
  store2zero(R + row * ld + c);
}

void launch_zero_upper_blocks(hipStream_t s, double *R, int64_t ld, int64_t npad) {
  if (npad <= PANEL) return;
  dim3 grid((unsigned)((npad / 2 + 255) / 256), (unsigned)(npad - PANEL), (unsigned)tl_batch.k);
  GOGP_KLAUNCH(zero_upper_kernel<double>, grid, dim3(256), 0, s, R, (long)ld, (long)npad,
                     tl_batch.stride);
}
void launch_zero_upper_blocks(hipStream_t s, float *R, int64_t ld, int64_t npad) {
  if (npad <= PANEL) return;
  dim3 grid((unsigned)((npad / 2 + 255) / 256), (unsigned)(npad - PANEL));
  GOGP_KLAUNCH(zero_upper_kernel<float>, grid, dim3(256), 0, s, R, (long)ld, (long)npad, 0L);
}

// Y[c0+i][c0+j] = Dinv[j][i]  (256x256 transpose of a diagonal-block inverse into
// the diagonal block of Y = L^-T)
template <class T>
__global__ __launch_bounds__(256) void ydiag_kernel(const T *__restrict__ Dinv,
                                                    T *__restrict__ Y, long ld, long bstride) {
  Dinv = cand(Dinv, bstride);  // candidate batching (common.h: Batch)
  Y = cand(Y, bstride);
  __shared__ T tile[32][33];
  const int bx = blockIdx.x & 7, by = blockIdx.x >> 3;  // 8x8 tiles of 32x32
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = 0; r < 32; r += 8) tile[ty + r][tx] = Dinv[(by * 32 + ty + r) * 256 + bx * 32 + tx];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) Y[(long)(bx * 32 + ty + r) * ld + by * 32 + tx] = tile[tx][ty + r];
}

void launch_ydiag(hipStream_t s, const double *Dinv, double *Ydiag, int64_t ld) {
  GOGP_KLAUNCH(ydiag_kernel<double>, dim3(64, 1, (unsigned)tl_batch.k), dim3(256), 0, s, Dinv, Ydiag,
                     (long)ld, tl_batch.stride);
}
void launch_ydiag(hipStream_t s, const float *Dinv, float *Ydiag, int64_t ld) {
  GOGP_KLAUNCH(ydiag_kernel<float>, dim3(64), dim3(256), 0, s, Dinv, Ydiag, (long)ld, 0L);
}

// ---- inverse of a super-panel's triangular diagonal block (api.hip: assemble_tinv) -------------------------------
// X (lower, row-major, leading dimension tld) = T^-1 of the nsub*256-wide block T = L[C0:CE, C0:CE] of the factor: with
// it Produce solves a whole super-panel of columns in ONE product (api.hip: produce_solve_t).  This kernel writes what
// needs no product -- the diagonal 256-blocks (Dinv_i) and zeros above them; the blocks below are written by
// blockmm_kernel.  (XT: optionally T^-T the same way; unused today.)  One workgroup per 32 x 32 tile.
template <class T>
__global__ __launch_bounds__(256) void tinv_init_kernel(const T *__restrict__ Dinv, T *__restrict__ X,
                                                        T *__restrict__ XT, int nsub, long tld, long bstride) {
  Dinv = cand(Dinv, bstride);  // candidate batching (common.h: Batch)
  X = cand(X, bstride);
  if (XT) XT = cand(XT, bstride);
  __shared__ T tile[32][33];
  const int tpr = nsub * 8;  // 32-tiles per row of the block
  const int bx = blockIdx.x % tpr, by = blockIdx.x / tpr;
  const int I = by >> 3, J = bx >> 3;  // 256-block row / column
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const long r0 = (long)by * 32, c0 = (long)bx * 32;
  if (I == J) {
    const T *D = Dinv + (long)I * 256 * 256 + (long)(by & 7) * 32 * 256 + (bx & 7) * 32;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
      const T v = D[(ty + r) * 256 + tx];
      X[(r0 + ty + r) * tld + c0 + tx] = v;
      tile[ty + r][tx] = v;
    }
    if (XT) {  // XT block (I, I) = Dinv_I^T: tile (by, bx) of Dinv lands at tile (bx, by)
      __syncthreads();
      const long tr0 = (long)I * 256 + (long)(bx & 7) * 32, tc0 = (long)I * 256 + (long)(by & 7) * 32;
#pragma unroll
      for (int r = 0; r < 32; r += 8) XT[(tr0 + ty + r) * tld + tc0 + tx] = tile[tx][ty + r];
    }
  } else if (I < J) {
#pragma unroll
    for (int r = 0; r < 32; r += 8) X[(r0 + ty + r) * tld + c0 + tx] = (T)0;
  } else if (XT) {
#pragma unroll
    for (int r = 0; r < 32; r += 8) XT[(r0 + ty + r) * tld + c0 + tx] = (T)0;
  }
}
void launch_tinv_init(hipStream_t s, const double *Dinv, double *X, double *XT, int nsub, int64_t tld) {
  GOGP_KLAUNCH(tinv_init_kernel<double>, dim3(nsub * 8 * nsub * 8, 1, (unsigned)tl_batch.k), dim3(256), 0, s, Dinv,
                     X, XT, nsub, (long)tld, tl_batch.stride);
}
void launch_tinv_init(hipStream_t s, const float *Dinv, float *X, float *XT, int nsub, int64_t tld) {
  GOGP_KLAUNCH(tinv_init_kernel<float>, dim3(nsub * 8 * nsub * 8), dim3(256), 0, s, Dinv, X, XT, nsub, (long)tld, 0L);
}

// Up to BLOCKMM_MAX products C_b (256 x 256) = alpha * A_b (256 x K_b) * B_b (K_b x 256) in ONE launch, all row-major
// (plain A B, which the A B^T tile kernel cannot do without transposed copies): the steps of the T^-1 assembly, tiny
// products whose cost is their launch, so that all blocks of one block diagonal go out together.  One
// workgroup per 64 x 64 tile of a product; K walked in chunks of 32 with the next chunk's global loads in flight
// behind the MFMAs of the current one; always fp64 arithmetic (v_mfma_f64_16x16x4_f64), T only on loads / stores.
constexpr int BLOCKMM_MAX = 6;
template <class T>
struct BlockMM {
  const T *A[BLOCKMM_MAX];
  const T *B[BLOCKMM_MAX];
  T *C[BLOCKMM_MAX];
  long lda[BLOCKMM_MAX], ldb[BLOCKMM_MAX], ldc[BLOCKMM_MAX];
  int K[BLOCKMM_MAX];
  double alpha;
  long bstride;
  int tshift;  // 64 x 64 tiles per side of a product: 1 << tshift (256 x 256: 2; 128 x 128: 1)
};
typedef double bmm_f64x4 __attribute__((ext_vector_type(4)));
template <class T>
__global__ __launch_bounds__(256, 2) void blockmm_kernel(BlockMM<T> g) {
  constexpr int KC = 32, AS = KC + 1, BS = 64 + 2;
  __shared__ double As[64 * AS];
  __shared__ double Bs[KC * BS];
  const int b = blockIdx.y;
  const T *A = cand(g.A[b], g.bstride);
  const T *B = cand(g.B[b], g.bstride);
  T *C = cand(g.C[b], g.bstride);
  const long lda = g.lda[b], ldb = g.ldb[b], ldc = g.ldc[b];
  const int K = g.K[b];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int tr = blockIdx.x >> g.tshift, tc = blockIdx.x & ((1 << g.tshift) - 1);  // 64 x 64 tile of the product
  A += (long)tr * 64 * lda;
  B += tc * 64;
  C += (long)tr * 64 * ldc + tc * 64;
  // staging maps: A chunk 64 x 32 (thread: row tid >> 2, 8 consecutive k from (tid & 3) * 8);
  //               B chunk 32 x 64 (thread: k-row tid >> 3, 8 consecutive columns from (tid & 7) * 8)
  const int ar = tid >> 2, ak = (tid & 3) * 8, bk = tid >> 3, bc = (tid & 7) * 8;
  double ra[8], rb[8];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ra[i] = (double)A[(long)ar * lda + k0 + ak + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = (double)B[(long)(k0 + bk) * ldb + bc + i];
  };
  bmm_f64x4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = (bmm_f64x4){0.0, 0.0, 0.0, 0.0};
  gload(0);
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();  // the previous chunk's readers are done
#pragma unroll
    for (int i = 0; i < 8; ++i) As[ar * AS + ak + i] = ra[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) Bs[bk * BS + bc + i] = rb[i];
    __syncthreads();
    if (k0 + KC < K) gload(k0 + KC);
    const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < KC / 4; ++kk) {
      const double a = As[(16 * w + fr) * AS + 4 * kk + fk];
#pragma unroll
      for (int n = 0; n < 4; ++n)
        acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bs[(4 * kk + fk) * BS + 16 * n + fr], acc[n], 0, 0, 0);
    }
  }
  // C fragment: column = lane & 15, row = (lane >> 4) + 4 v
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int v = 0; v < 4; ++v)
      C[(long)(16 * w + (lane >> 4) + 4 * v) * ldc + 16 * n + (lane & 15)] = (T)(g.alpha * acc[n][v]);
}
template <class T>
static void launch_blockmm_t(hipStream_t s, int nprod, const T *const *A, const int64_t *lda, const T *const *B,
                             const int64_t *ldb, T *const *C, const int64_t *ldc, const int *K, double alpha,
                             unsigned nz, long bstride, int side = 256) {
  if (nprod <= 0) return;
  BlockMM<T> g;
  for (int b = 0; b < nprod; ++b) {
    g.A[b] = A[b];
    g.B[b] = B[b];
    g.C[b] = C[b];
    g.lda[b] = lda[b];
    g.ldb[b] = ldb[b];
    g.ldc[b] = ldc[b];
    g.K[b] = K[b];
  }
  g.alpha = alpha;
  g.bstride = bstride;
  g.tshift = side == 128 ? 1 : 2;
  GOGP_KLAUNCH(blockmm_kernel<T>, dim3(1u << (2 * g.tshift), (unsigned)nprod, nz), dim3(256), 0, s, g);
}
void launch_blockmm(hipStream_t s, int nprod, const double *const *A, const int64_t *lda, const double *const *B,
                    const int64_t *ldb, double *const *C, const int64_t *ldc, const int *K, double alpha, int side) {
  launch_blockmm_t<double>(s, nprod, A, lda, B, ldb, C, ldc, K, alpha, (unsigned)tl_batch.k, tl_batch.stride, side);
}
void launch_blockmm(hipStream_t s, int nprod, const float *const *A, const int64_t *lda, const float *const *B,
                    const int64_t *ldb, float *const *C, const int64_t *ldc, const int *K, double alpha) {
  launch_blockmm_t<float>(s, nprod, A, lda, B, ldb, C, ldc, K, alpha, 1u, 0L);
}

// zero a rows x cols block (cols a multiple of 2, 16-B aligned)
template <class T>
__global__ __launch_bounds__(256) void zero_block_kernel(T *__restrict__ B, long ld, long cols,
                                                         long bstride) {
  B = cand(B, bstride);  // candidate batching (common.h: Batch)
  const long r = blockIdx.y;
  const long c = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (c >= cols) return;
  store2zero(B + r * ld + c);
}

void launch_zero_block(hipStream_t s, double *B, int64_t ld, int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return;
  dim3 grid((unsigned)((cols / 2 + 255) / 256), (unsigned)rows, (unsigned)tl_batch.k);
  GOGP_KLAUNCH(zero_block_kernel<double>, grid, dim3(256), 0, s, B, (long)ld, (long)cols,
                     tl_batch.stride);
}
void launch_zero_block(hipStream_t s, float *B, int64_t ld, int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return;
  dim3 grid((unsigned)((cols / 2 + 255) / 256), (unsigned)rows);
  GOGP_KLAUNCH(zero_block_kernel<float>, grid, dim3(256), 0, s, B, (long)ld, (long)cols, 0L);
}

__global__ void fill_kernel(double *p, long count, double v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long)gridDim.x * blockDim.x)
    p[i] = v;
}

__global__ void axpy_kernel(double *__restrict__ a, const double *__restrict__ b, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) a[i] += b[i];
}
void launch_axpy(hipStream_t s, double *a, const double *b, int64_t count) {
  if (count <= 0) return;
  GOGP_KLAUNCH(axpy_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, a, b, (long)count);
}

// out[0] = sum_{i<n} a_i b_i  (single workgroup, fixed order)
__global__ __launch_bounds__(1024) void dot_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                   long n, double *__restrict__ out) {
  __shared__ double red[16];
  double v = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) v += a[i] * b[i];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = t;
  }
}
void launch_dot(hipStream_t s, const double *a, const double *b, int64_t n, double *out) {
  GOGP_KLAUNCH(dot_kernel, dim3(1), dim3(1024), 0, s, a, b, (long)n, out);
}

void launch_fill(hipStream_t s, double *p, int64_t count, double v) {
  if (count <= 0) return;
  int blocks = (int)((count + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  GOGP_KLAUNCH(fill_kernel, dim3(blocks), dim3(256), 0, s, p, (long)count, v);
}

template <class T>
__global__ void extract_lower_kernel(const T *__restrict__ L, long ld, long n,
                                     double *__restrict__ out) {
  const long i = blockIdx.y;
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  out[i * n + j] = (j <= i) ? (double)L[i * ld + j] : 0.0;
}

void launch_extract_lower(hipStream_t s, const double *L, int64_t ld, int64_t n, double *out) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
  GOGP_KLAUNCH(extract_lower_kernel<double>, grid, dim3(256), 0, s, L, (long)ld, (long)n, out);
}
void launch_extract_lower(hipStream_t s, const float *L, int64_t ld, int64_t n, double *out) {
  if (n <= 0) return;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)n);
  GOGP_KLAUNCH(extract_lower_kernel<float>, grid, dim3(256), 0, s, L, (long)ld, (long)n, out);
}

// rows x cols block conversions between the fp32 matrices and the fp64 scratch of the
// diagonal-block kernel (fp32 path: the diagonal blocks are factored and inverted in fp64)
template <class S, class D>
__global__ __launch_bounds__(256) void convert_block_kernel(const S *__restrict__ src, long lds_,
                                                            D *__restrict__ dst, long ldd, int cols) {
  const long r = blockIdx.x;
  for (int c = threadIdx.x; c < cols; c += 256) dst[r * ldd + c] = (D)src[r * lds_ + c];
}
void launch_convert_block(hipStream_t s, const float *src, int64_t lds_, double *dst, int64_t ldd,
                          int rows, int cols) {
  GOGP_KLAUNCH((convert_block_kernel<float, double>), dim3(rows), dim3(256), 0, s, src, (long)lds_,
                     dst, (long)ldd, cols);
}
void launch_convert_block(hipStream_t s, const double *src, int64_t lds_, float *dst, int64_t ldd,
                          int rows, int cols) {
  GOGP_KLAUNCH((convert_block_kernel<double, float>), dim3(rows), dim3(256), 0, s, src, (long)lds_,
                     dst, (long)ldd, cols);
}
void launch_convert_block(hipStream_t s, const double *src, int64_t lds_, double *dst, int64_t ldd,
                          int rows, int cols) {  // same type: a strided block copy
  GOGP_KLAUNCH((convert_block_kernel<double, double>), dim3(rows), dim3(256), 0, s, src, (long)lds_,
                     dst, (long)ldd, cols);
}

// pack a dense n x n lower factor into the padded buffer (identity padding)
template <class T>
__global__ void pack_lower_kernel(const double *__restrict__ in, long n, long npad,
                                  T *__restrict__ L, long ld) {
  const long i = blockIdx.y;
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npad) return;
  double v = 0.0;
  if (i < n && j < n) v = (j <= i) ? in[i * n + j] : 0.0;
  else if (i == j) v = 1.0;
  L[i * ld + j] = (T)v;
}

void launch_pack_lower(hipStream_t s, const double *in, int64_t n, int64_t npad, double *L,
                       int64_t ld) {
  dim3 grid((unsigned)((npad + 255) / 256), (unsigned)npad);
  GOGP_KLAUNCH(pack_lower_kernel<double>, grid, dim3(256), 0, s, in, (long)n, (long)npad, L,
                     (long)ld);
}
void launch_pack_lower(hipStream_t s, const double *in, int64_t n, int64_t npad, float *L,
                       int64_t ld) {
  dim3 grid((unsigned)((npad + 255) / 256), (unsigned)npad);
  GOGP_KLAUNCH(pack_lower_kernel<float>, grid, dim3(256), 0, s, in, (long)n, (long)npad, L,
                     (long)ld);
}

// sigma_j = sqrt(prior_j - q_j), unclamped (gp/gp.go:356)
__global__ void sigma_kernel(const double *__restrict__ prior, const double *__restrict__ q,
                             long m, double *__restrict__ sigma) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < m) sigma[j] = sqrt(prior[j] - (q ? q[j] : 0.0));
}

void launch_sigma(hipStream_t s, const double *prior, const double *q, int64_t m,
                  double *sigma) {
  if (m <= 0) return;
  GOGP_KLAUNCH(sigma_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, prior, q,
                     (long)m, sigma);
}

// ---- helpers of the sharded (2-D block-cyclic) evaluation ------------------------------------

// dst (n x n, ldd) = src (n x n, lds)^T, n a multiple of 32
template <class T>
__global__ __launch_bounds__(256) void transpose_sq_kernel(const T *__restrict__ src, long lds_,
                                                           T *__restrict__ dst, long ldd, int nt) {
  __shared__ T tile[32][33];
  const int bx = blockIdx.x % nt, by = blockIdx.x / nt;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 32; r += 8) tile[ty + r][tx] = src[(long)(by * 32 + ty + r) * lds_ + bx * 32 + tx];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) dst[(long)(bx * 32 + ty + r) * ldd + by * 32 + tx] = tile[tx][ty + r];
}

void launch_transpose_sq(hipStream_t s, const double *src, int64_t lds_, double *dst, int64_t ldd,
                         int n) {
  const int nt = n / 32;
  GOGP_KLAUNCH(transpose_sq_kernel<double>, dim3(nt * nt), dim3(256), 0, s, src, (long)lds_, dst,
                     (long)ldd, nt);
}
void launch_transpose_sq(hipStream_t s, const float *src, int64_t lds_, float *dst, int64_t ldd, int n) {
  const int nt = n / 32;
  GOGP_KLAUNCH(transpose_sq_kernel<float>, dim3(nt * nt), dim3(256), 0, s, src, (long)lds_, dst,
                     (long)ldd, nt);
}

// dst block d (blk doubles, contiguous) = src block (first + d * stride), d < nblk
__global__ __launch_bounds__(256) void pack_blocks_kernel(double *__restrict__ dst,
                                                          const double *__restrict__ src, long blk,
                                                          int first, int stride) {
  const long d = blockIdx.y;
  const double2 *sp = reinterpret_cast<const double2 *>(src + (long)(first + d * stride) * blk);
  double2 *dp = reinterpret_cast<double2 *>(dst + d * blk);
  const long n2 = blk / 2, step = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  // four 16-B loads in flight per thread before the stores
  for (; i + 3 * step < n2; i += 4 * step) {
    const double2 a = sp[i], b = sp[i + step], c = sp[i + 2 * step], e = sp[i + 3 * step];
    dp[i] = a;
    dp[i + step] = b;
    dp[i + 2 * step] = c;
    dp[i + 3 * step] = e;
  }
  for (; i < n2; i += step) dp[i] = sp[i];
}

void launch_pack_blocks(hipStream_t s, double *dst, const double *src, int nblk, int64_t blk,
                        int first, int stride) {
  if (nblk <= 0) return;
  GOGP_KLAUNCH(pack_blocks_kernel, dim3(64, (unsigned)nblk), dim3(256), 0, s, dst, src,
                     (long)blk, first, stride);
}

// out[c] = sum_{r < rows} chunk[r][c] * v[r]   (chunk rows x nb, leading dimension nb).
// Two stages, fixed summation order: workgroup (column group of 64, row slab of TDOT_SLAB rows)
// writes part[slab][c]; the finish kernel adds the slabs.  part: (rows/TDOT_SLAB + 1) * nb doubles.
constexpr int TDOT_SLAB = 256;
template <class T>
__global__ __launch_bounds__(256) void chunk_tdot_kernel(const T *__restrict__ chunk, long rows,
                                                         int nb, const double *__restrict__ v,
                                                         double *__restrict__ part) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long r0 = (long)blockIdx.y * TDOT_SLAB;
  const long r1 = (r0 + TDOT_SLAB < rows) ? r0 + TDOT_SLAB : rows;
  double a0 = 0.0, a1 = 0.0;
  long r = r0 + wid;
  for (; r + 4 < r1; r += 8) {
    a0 += (double)chunk[r * nb + c] * v[r];
    a1 += (double)chunk[(r + 4) * nb + c] * v[r + 4];
  }
  if (r < r1) a0 += (double)chunk[r * nb + c] * v[r];
  red[wid][lane] = a0 + a1;
  __syncthreads();
  if (wid == 0) part[(long)blockIdx.y * nb + c] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
__global__ void chunk_tdot_finish_kernel(const double *__restrict__ part, int nslab, int nb,
                                         double *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nb) return;
  double s = 0.0;
  for (int q = 0; q < nslab; ++q) s += part[(long)q * nb + c];
  out[c] = s;
}

int64_t chunk_tdot_scratch(int64_t max_rows, int nb) { return (max_rows / TDOT_SLAB + 1) * (int64_t)nb; }

template <class T>
static void chunk_tdot_t(hipStream_t s, const T *chunk, int64_t rows, int nb, const double *v, double *part,
                         double *out) {
  const int nslab = (int)((rows + TDOT_SLAB - 1) / TDOT_SLAB);
  GOGP_KLAUNCH(chunk_tdot_kernel<T>, dim3(nb / 64, nslab), dim3(256), 0, s, chunk, (long)rows, nb, v, part);
  GOGP_KLAUNCH(chunk_tdot_finish_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, part, nslab, nb, out);
}
void launch_chunk_tdot(hipStream_t s, const double *chunk, int64_t rows, int nb, const double *v,
                       double *part, double *out) {
  chunk_tdot_t(s, chunk, rows, nb, v, part, out);
}
void launch_chunk_tdot(hipStream_t s, const float *chunk, int64_t rows, int nb, const double *v,
                       double *part, double *out) {
  chunk_tdot_t(s, chunk, rows, nb, v, part, out);
}

// alpha partial of one rank: for every local row (local row block bi, global block
// gI = bi*Pr + pr) out[global row] = sum over the local chunks bj with gP = bj*Pc + pc >= gI of
// sum_c Ych[bj][bi][r][c] * z[gP*nb + c].  Ych: nloc chunks of (mloc*nb) x nb.  One wave per row.
template <class T>
__global__ __launch_bounds__(256) void chunk_alpha_kernel(const T *__restrict__ Ych, int mloc,
                                                          int nloc, int nb, BlockMap map,
                                                          const double *__restrict__ z,
                                                          double *__restrict__ out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long lrow = (long)blockIdx.x * 4 + wid;
  if (lrow >= (long)mloc * nb) return;
  const int bi = (int)(lrow / nb);
  const int gI = bi * map.Pr + map.pr;
  const long chunk_sz = (long)mloc * nb * nb;
  double s0 = 0.0, s1 = 0.0;
  for (int bj = 0; bj < nloc; ++bj) {
    const int gP = bj * map.Pc + map.pc;
    if (gP < gI) continue;
    const T *row = Ych + bj * chunk_sz + lrow * nb;
    const double *zz = z + (long)gP * nb;
    for (int c = lane * 2; c < nb; c += 128) {
      double ya, yb;
      load2(row + c, ya, yb);
      const double2 q = *reinterpret_cast<const double2 *>(zz + c);
      s0 += ya * q.x;
      s1 += yb * q.y;
    }
  }
  double sum = s0 + s1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) out[map.grow(lrow)] = sum;
}

template <class T>
static void chunk_alpha_t(hipStream_t s, const T *Ych, int mloc, int nloc, int nb, BlockMap map, const double *z,
                          double *out) {
  const long rows = (long)mloc * nb;
  if (rows <= 0) return;
  GOGP_KLAUNCH(chunk_alpha_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, Ych, mloc,
                     nloc, nb, map, z, out);
}
void launch_chunk_alpha(hipStream_t s, const double *Ych, int mloc, int nloc, int nb, BlockMap map,
                        const double *z, double *out) {
  chunk_alpha_t(s, Ych, mloc, nloc, nb, map, z, out);
}
void launch_chunk_alpha(hipStream_t s, const float *Ych, int mloc, int nloc, int nb, BlockMap map,
                        const double *z, double *out) {
  chunk_alpha_t(s, Ych, mloc, nloc, nb, map, z, out);
}

// |Y_local|_F^2 of one rank over its chunks of Y = L^-T (rows < n), in fp64: part[local row] = sum over the local chunks
// bj with gP >= gI of sum_c Ych[bj][bi][r][c]^2 (the loop of chunk_alpha_kernel), then one workgroup adds the rows in a
// fixed order.  Summed over the ranks it is tr(K^-1) -- the float shards' gradient takes tr(alpha alpha^T - K^-1) from
// it instead of from the float diagonal of K^-1 (api.hip: fp32_gradient_identities).
template <class T>
__global__ __launch_bounds__(256) void chunk_rowsq_kernel(const T *__restrict__ Ych, int mloc, int nloc, int nb,
                                                          BlockMap map, long n, double *__restrict__ part) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long lrow = (long)blockIdx.x * 4 + wid;
  if (lrow >= (long)mloc * nb) return;
  const int bi = (int)(lrow / nb);
  const int gI = bi * map.Pr + map.pr;
  const long chunk_sz = (long)mloc * nb * nb;
  double s0 = 0.0, s1 = 0.0;
  if (map.grow(lrow) < n)
    for (int bj = 0; bj < nloc; ++bj) {
      const int gP = bj * map.Pc + map.pc;
      if (gP < gI) continue;
      const T *row = Ych + bj * chunk_sz + lrow * nb;
      for (int c = lane * 2; c < nb; c += 128) {
        double ya, yb;
        load2(row + c, ya, yb);
        s0 += ya * ya;
        s1 += yb * yb;
      }
    }
  double sum = s0 + s1;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) part[lrow] = sum;
}
template <class T>
static void chunk_sumsq_t(hipStream_t s, const T *Ych, int mloc, int nloc, int nb, BlockMap map, int64_t n, double *part,
                          double *out) {
  const long rows = (long)mloc * nb;
  if (rows <= 0) return;
  GOGP_KLAUNCH(chunk_rowsq_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, Ych, mloc, nloc, nb, map, (long)n,
               part);
  GOGP_KLAUNCH(sum_kernel, dim3(1), dim3(1024), 0, s, (const double *)part, rows, out);
}
void launch_chunk_sumsq(hipStream_t s, const double *Ych, int mloc, int nloc, int nb, BlockMap map, int64_t n, double *part,
                        double *out) {
  chunk_sumsq_t(s, Ych, mloc, nloc, nb, map, n, part, out);
}
void launch_chunk_sumsq(hipStream_t s, const float *Ych, int mloc, int nloc, int nb, BlockMap map, int64_t n, double *part,
                        double *out) {
  chunk_sumsq_t(s, Ych, mloc, nloc, nb, map, n, part, out);
}

// acc[0] += sum_{i < nb, row0 + i < n} 2 log L[i][i]   (one diagonal block; single workgroup)
__global__ __launch_bounds__(256) void logdet_block_kernel(const double *__restrict__ L, long ld,
                                                           long row0, long n, int nb,
                                                           double *__restrict__ acc) {
  __shared__ double red[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256)
    if (row0 + i < n) a += 2.0 * log(L[(long)i * ld + i]);
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) acc[0] += red[0] + red[1] + red[2] + red[3];
}

void launch_logdet_block(hipStream_t s, const double *L, int64_t ld, int64_t row0, int64_t n, int nb,
                         double *acc) {
  GOGP_KLAUNCH(logdet_block_kernel, dim3(1), dim3(256), 0, s, L, (long)ld, (long)row0, (long)n,
                     nb, acc);
}

// out[0] = sum_{i<n} z_i^2 ; out[1] = (double)*info   (single workgroup, fixed order)
__global__ __launch_bounds__(1024) void sumsq_info_kernel(const double *__restrict__ z, long n,
                                                          const long long *__restrict__ info,
                                                          double *__restrict__ out) {
  __shared__ double red[16];
  double a = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) a += z[i] * z[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    out[0] = t;
    if (info) out[1] = (double)*info;
  }
}

void launch_sumsq_info(hipStream_t s, const double *z, int64_t n, const long long *info, double *out) {
  GOGP_KLAUNCH(sumsq_info_kernel, dim3(1), dim3(1024), 0, s, z, (long)n, info, out);
}

__global__ void info_to_double_kernel(const long long *__restrict__ info, double *__restrict__ out) {
  out[0] = (double)*info;
}

void launch_info_to_double(hipStream_t s, const long long *info, double *out) {
  GOGP_KLAUNCH(info_to_double_kernel, dim3(1), dim3(1), 0, s, info, out);
}

}  // namespace gogp
