// diag.hip -- Cholesky factor and inverse of one 128x128 diagonal block, one
// workgroup of 1024 threads (16 waves).
//
// Reference counterpart: the unblocked core of gonum's Dpotrf behind
// mat.Cholesky.Factorize (gp/gp.go:228) -- Factorize reports "not positive
// definite" by returning false, here by writing the failing pivot to *info.
//
// Phase 1 (potrf, right-looking, register-resident): thread (ty,tx) of a 32x32
// grid owns elements (ty+32a, tx+32b), a,b in 0..3.  Step j publishes column j
// through a double-buffered 128-entry LDS vector (ONE barrier per step) and
// applies  a_ic -= a_ij a_cj / a_jj  in registers; the scaling by 1/sqrt(a_jj)
// is deferred to the end (L_ic = a_ic^(c) / sqrt(a_cc^(c))).
// Phase 2 (inverse of L, wave-synchronous): column j of X = L^-1 is owned by 8
// lanes of one wave; lane s keeps x_qj for q = s mod 8 in registers, the dot
// product of forward substitution is split 8 ways and combined with 3 xor
// shuffles -- no workgroup barrier inside the 128-row sweep.
#include "common.h"

namespace gogp {

constexpr int DB = 128;
constexpr int DLD = DB + 1;

template <bool DO_POTRF>
__global__ __launch_bounds__(1024) void diag128_kernel(const double *__restrict__ A, long ld,
                                                       double *__restrict__ Lout, long ldl,
                                                       double *__restrict__ Dinv, long row0,
                                                       long nvalid, long long *info) {
  __shared__ double S[DB * DLD];
  __shared__ double colbuf[2][DB];
  __shared__ double dsq[DB];   // sqrt of the pivots
  __shared__ double dinv[DB];  // 1 / L_ii
  const int tid = threadIdx.x;

  if (DO_POTRF) {
    const int tx = tid & 31, ty = tid >> 5;
    double e[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = ty + 32 * a, c = tx + 32 * b;
        e[a][b] = (c <= i) ? A[(long)i * ld + c] : 0.0;
      }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
      for (int jx = 0; jx < 32; ++jx) {
        const int j = jb * 32 + jx;
        double *cb = colbuf[j & 1];
        if (tx == jx) {
#pragma unroll
          for (int a = 0; a < 4; ++a) cb[ty + 32 * a] = e[a][jb];
        }
        __syncthreads();
        double d = cb[j];
        if (!(d > 0.0)) {
          if (tid == 0 && row0 + j < nvalid) {
            // first failure wins (stream order makes earlier blocks earlier)
            if (*info == 0) *info = (long long)(row0 + j + 1);
          }
          d = 1.0;
        }
        if (tid == 0) dsq[j] = sqrt(d);
        const double invd = 1.0 / d;
        double ci[4], cc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) ci[a] = cb[ty + 32 * a];
#pragma unroll
        for (int b = 0; b < 4; ++b) cc[b] = cb[tx + 32 * b] * invd;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if (b < jb) continue;
            const int i = ty + 32 * a, c = tx + 32 * b;
            if (c > j && c <= i) e[a][b] -= ci[a] * cc[b];
          }
      }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = ty + 32 * a, c = tx + 32 * b;
        S[i * DLD + c] = (c <= i) ? e[a][b] / dsq[c] : 0.0;
      }
    __syncthreads();
    // factor out (upper triangle zero-filled)
    for (int idx = tid; idx < DB * DB; idx += 1024) {
      const int i = idx >> 7, c = idx & 127;
      Lout[(long)i * ldl + c] = S[i * DLD + c];
    }
  } else {
    for (int idx = tid; idx < DB * DB; idx += 1024) {
      const int i = idx >> 7, c = idx & 127;
      S[i * DLD + c] = (c <= i) ? A[(long)i * ld + c] : 0.0;
    }
    __syncthreads();
  }
  if (tid < DB) dinv[tid] = 1.0 / S[tid * DLD + tid];
  __syncthreads();

  // ---- phase 2: X = L^-1, column j = tid>>3, slice s = tid&7 -----------------
  {
    const int j = tid >> 3, s = tid & 7;
    const int jmin = (tid >> 6) << 3;  // first column of this wave
    double xr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) xr[u] = 0.0;
#pragma unroll
    for (int ub = 0; ub < 16; ++ub) {
      // wave-uniform skip of 8-row groups above every column of this wave
      if (8 * ub + 8 > jmin) {
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
          const int i = 8 * ub + ii;
          double part = 0.0;
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            if (u <= ub) {  // q = 8u+s < i  needs  u <= ub
              const int q = 8 * u + s;
              const double l = (q < i) ? S[i * DLD + q] : 0.0;
              part += l * xr[u];
            }
          }
          part += __shfl_xor(part, 1);
          part += __shfl_xor(part, 2);
          part += __shfl_xor(part, 4);
          const double xi = ((i == j ? 1.0 : 0.0) - part) * dinv[i];
          if (ii == s) xr[ub] = (i >= j) ? xi : 0.0;
        }
      }
    }
    __syncthreads();  // every wave is done reading L from S
#pragma unroll
    for (int u = 0; u < 16; ++u) S[(8 * u + s) * DLD + j] = xr[u];
    __syncthreads();
    for (int idx = tid; idx < DB * DB; idx += 1024) {
      const int i = idx >> 7, c = idx & 127;
      Dinv[idx] = S[i * DLD + c];
    }
  }
}

void launch_diag128(hipStream_t s, const double *A, int64_t ld, double *Lout, int64_t ldl,
                    double *Dinv, int64_t row0, int64_t nvalid, long long *info) {
  hipLaunchKernelGGL(diag128_kernel<true>, dim3(1), dim3(1024), 0, s, A, (long)ld, Lout,
                     (long)ldl, Dinv, (long)row0, (long)nvalid, info);
}

void launch_diag128_inv_only(hipStream_t s, const double *L, int64_t ld, double *Dinv) {
  hipLaunchKernelGGL(diag128_kernel<false>, dim3(1), dim3(1024), 0, s, L, (long)ld,
                     (double *)nullptr, 0L, Dinv, 0L, 0L, (long long *)nullptr);
}

}  // namespace gogp
