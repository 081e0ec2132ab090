// gram.hip -- O(N^2) pairwise kernel evaluation (HBM-write-bound).
//
//   gram_kernel<false>: lower-triangular 64x64 tiles of
//       K_ij = Simil(x_i, x_j) + [i==j] Noise          gp/gp.go:109-156,220-225
//     (the reference walks j >= i and mirrors with SetSym; the blocked lower
//      Cholesky that follows only reads the lower triangle, so only that is
//      written).  Rows/columns >= n are padding: identity.
//   gram_kernel<true>: cross-covariance for Produce, stored transposed,
//       KsT[j][i] = Simil(x_i, z_j)                    gp/gp.go:322-332
//   prior_kernel: k(z_j, z_j)                          gp/gp.go:269-278
//
// Layout: one workgroup = 64x64 tile, 256 threads.  The row inputs are staged
// in LDS row-major (read as wave-uniform broadcasts), the column inputs
// transposed [d][col] so that the 64 lanes of a wave read consecutive
// addresses; each wave writes 512 contiguous bytes per output row.
#include <algorithm>

#include "kern_eval.h"

namespace gogp {

// T: element type of the output matrix (double, or float on the fp32 path: the kernel value
// is computed in fp64 from the fp64 inputs and rounded once on store)
template <bool CROSS, class T>
__global__ __launch_bounds__(256) void gram_kernel(const DevParams *__restrict__ Pp,
                                                   const double *__restrict__ Rsrc, long nrows,
                                                   const double *__restrict__ Csrc, long ncols,
                                                   T *__restrict__ Out, long ld, int ntc,
                                                   int strip_w, int toff, long bstride) {
  extern __shared__ double sm[];
  const DevParams &P = *cand(Pp, bstride);  // candidate batching: parameters and output per candidate
  Out = cand(Out, bstride);
  const int D = P.ndim;
  double *Ri = sm;            // [64][D]
  double *CjT = sm + 64 * D;  // [D][64]
  const int tid = threadIdx.x;
  int ti, tj;
  if (CROSS) {
    ti = blockIdx.x / ntc;
    tj = blockIdx.x - ti * ntc;
  } else if (strip_w > 0) {
    // the first strip_w tile columns of the lower triangle (all rows; the few tiles above
    // the diagonal exit)
    ti = blockIdx.x / strip_w;
    tj = blockIdx.x - ti * strip_w;
    if (ti < tj) return;  // whole-workgroup exit
  } else {
    // lower triangle of the tile grid shifted by toff tile rows / columns
    const int t = blockIdx.x;
    ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    tj = t - ti * (ti + 1) / 2;
    ti += toff;
    tj += toff;
  }
  const long r0 = (long)ti * 64, c0 = (long)tj * 64;
  for (int idx = tid; idx < 64 * D; idx += 256) {
    const int r = idx / D, d = idx - r * D;
    Ri[idx] = (r0 + r < nrows) ? Rsrc[(r0 + r) * D + d] : 0.0;
  }
  // the transposed copy with the LANES along a row of CjT (consecutive LDS addresses): with the lanes along d the
  // stores of a wave hit addresses 512 B apart -- one bank, D-way conflicts: SQ_LDS_BANK_CONFLICT was 64 % of this
  // kernel's LDS cycles (round-3 PMC pass).  The strided global reads cost nothing: the tile's 64 x D inputs are one
  // or two L2-resident lines per lane group.
  for (int idx = tid; idx < 64 * D; idx += 256) {
    const int d = idx >> 6, r = idx & 63;
    CjT[idx] = (c0 + r < ncols) ? Csrc[(c0 + r) * D + d] : 0.0;
  }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;
  const long gj = c0 + tx;
  const double *cj = CjT + tx;
  if (!CROSS && P.nterms == 1 && P.kind[0] != GOGP_K_PERIODIC) {
    // One radial term (every BASELINE configuration): dimension loop outside, the thread's 16 rows
    // inside.  The column coordinate is read from LDS once per dimension instead of once per
    // (row, dimension), and the rows' coordinates are wave-uniform: they come straight from X
    // through scalar loads (X is padded to npad rows), not through LDS.  Same operations in the same
    // order per pair as simil_value() -- bit-identical -- at 1/32 of its LDS reads.
    const int rbase = __builtin_amdgcn_readfirstlane(ty) * 16;
    const double *xr = Rsrc + (r0 + rbase) * D;
    double s[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) s[rr] = 0.0;
    for (int d = 0; d < D; ++d) {
      const double il = P.inv_len[0][d];
      const double c = cj[d * 64];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const double u = (xr[rr * D + d] - c) * il;
        s[rr] += u * u;
      }
    }
    const int kind = P.kind[0];
    const double c0s = P.c[0];
#pragma unroll 2
    for (int rr = 0; rr < 16; ++rr) {
      const long gi = r0 + rbase + rr;
      double k;
      if (gi < nrows && gj < ncols) {
        double f, dfdr2;
        radial_eval(kind, s[rr], f, dfdr2);
        k = 0.0 + c0s * f;
        if (gi == gj) k += P.noise_var;
      } else {
        k = (gi == gj) ? 1.0 : 0.0;
      }
      Out[gi * ld + gj] = (T)k;
    }
    return;
  }
#pragma unroll 2
  for (int rr = 0; rr < 16; ++rr) {
    const int r = ty * 16 + rr;
    const long gi = r0 + r;
    const double *ri = Ri + r * D;
    double k;
    if (gi < nrows && gj < ncols) {
      k = simil_value(
          P, [&](int d) { return ri[d]; }, [&](int d) { return cj[d * 64]; });
      if (!CROSS && gi == gj) k += P.noise_var;
    } else {
      k = (!CROSS && gi == gj) ? 1.0 : 0.0;
    }
    Out[gi * ld + gj] = (T)k;
  }
}

// 2-D block-cyclic variant: rectangular grid over the LOCAL 64x64 tiles, global indices
// through the block map (see common.h).
template <class T>
__global__ __launch_bounds__(256) void gram_local_kernel(const DevParams *__restrict__ Pp,
                                                         const double *__restrict__ X, long n,
                                                         T *__restrict__ Out, long ld, int ntc,
                                                         BlockMap map) {
  extern __shared__ double sm[];
  const DevParams &P = *Pp;
  const int D = P.ndim;
  double *Ri = sm;
  double *CjT = sm + 64 * D;
  const int tid = threadIdx.x;
  const int ti = blockIdx.x / ntc, tj = blockIdx.x - ti * ntc;
  const long lr0 = (long)ti * 64, lc0 = (long)tj * 64;
  const long r0 = map.grow(lr0), c0 = map.gcol(lc0);
  const int tx = tid & 63, ty = tid >> 6;
  if ((r0 >> map.nb_shift) < (c0 >> map.nb_shift)) {  // strictly upper distribution block: R := 0
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) Out[(lr0 + ty * 16 + rr) * ld + lc0 + tx] = (T)0;
    return;
  }
  if (r0 + 63 < c0) return;  // upper tile inside a diagonal block: never read
  for (int idx = tid; idx < 64 * D; idx += 256) {
    const int r = idx / D, d = idx - r * D;
    Ri[idx] = (r0 + r < n) ? X[(r0 + r) * D + d] : 0.0;
  }
  for (int idx = tid; idx < 64 * D; idx += 256) {  // lanes along a row of CjT: conflict-free stores (gram_kernel)
    const int d = idx >> 6, r = idx & 63;
    CjT[idx] = (c0 + r < n) ? X[(c0 + r) * D + d] : 0.0;
  }
  __syncthreads();
  const long gj = c0 + tx;
  const double *cj = CjT + tx;
  if (P.nterms == 1 && P.kind[0] != GOGP_K_PERIODIC) {  // one radial term: as in gram_kernel
    const int rbase = __builtin_amdgcn_readfirstlane(ty) * 16;
    const double *xr = X + (r0 + rbase) * D;
    double s[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) s[rr] = 0.0;
    for (int d = 0; d < D; ++d) {
      const double il = P.inv_len[0][d];
      const double c = cj[d * 64];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const double u = (xr[rr * D + d] - c) * il;
        s[rr] += u * u;
      }
    }
    const int kind = P.kind[0];
    const double c0s = P.c[0];
#pragma unroll 2
    for (int rr = 0; rr < 16; ++rr) {
      const int r = rbase + rr;
      const long gi = r0 + r;
      double k;
      if (gi < n && gj < n) {
        double f, dfdr2;
        radial_eval(kind, s[rr], f, dfdr2);
        k = 0.0 + c0s * f;
        if (gi == gj) k += P.noise_var;
      } else {
        k = (gi == gj) ? 1.0 : 0.0;
      }
      Out[(lr0 + r) * ld + lc0 + tx] = (T)k;
    }
    return;
  }
#pragma unroll 2
  for (int rr = 0; rr < 16; ++rr) {
    const int r = ty * 16 + rr;
    const long gi = r0 + r;
    const double *ri = Ri + r * D;
    double k;
    if (gi < n && gj < n) {
      k = simil_value(
          P, [&](int d) { return ri[d]; }, [&](int d) { return cj[d * 64]; });
      if (gi == gj) k += P.noise_var;
    } else {
      k = (gi == gj) ? 1.0 : 0.0;
    }
    Out[(lr0 + r) * ld + lc0 + tx] = (T)k;
  }
}

// Residual of the linear system with the EXACT (fp64, recomputed on the fly) Gram matrix:
//   part[slab][i] = sum_{j in slab} K_ij v_j     (K_ii includes the noise variance)
// for the iterative refinement of alpha on the fp32 path: K is never read back from its rounded
// fp32 copy.  One workgroup = 64 rows x one slab of column tiles; the slabs are summed (fixed
// order) by kmatvec_finish_kernel, which also forms r = y - K v.
template <bool RADIAL1>
__global__ __launch_bounds__(256) void kmatvec_kernel(const DevParams *__restrict__ Pp,
                                                      const double *__restrict__ X, long n,
                                                      const double *__restrict__ v, long npad,
                                                      int tiles_per_slab, double *__restrict__ part,
                                                      int tile_begin, int tile_end) {
  extern __shared__ double sm[];
  const DevParams &P = *Pp;
  const int D = P.ndim;
  double *RiT = sm;             // [D][64] rows of this workgroup, transposed: lane tx reads
                                // RiT[d*64 + tx] -- consecutive addresses, conflict-free
  double *CjT = sm + 64 * D;    // [D][64] current column tile (read as broadcasts)
  double *vj = CjT + 64 * D;    // [64]
  double *red = vj + 64;        // [4][64]
  const int tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;
  const long r0 = (long)blockIdx.x * 64;
  const int slab = blockIdx.y;
  for (int idx = tid; idx < 64 * D; idx += 256) {  // lanes along a row of RiT: conflict-free stores (gram_kernel)
    const int d = idx >> 6, r = idx & 63;
    RiT[idx] = (r0 + r < n) ? X[(r0 + r) * D + d] : 0.0;
  }
  // thread (tx, ty): row r0 + tx, columns ty*16 .. ty*16+15 of every tile
  double acc = 0.0;
  const long gi = r0 + tx;
  // column tiles [tile_begin, tile_end): all of them, or one rank's share of a sharded evaluation
  for (int t = tile_begin + slab * tiles_per_slab; t < tile_begin + (slab + 1) * tiles_per_slab && t < tile_end;
       ++t) {
    const long c0 = (long)t * 64;
    __syncthreads();
    if (!RADIAL1)
      for (int idx = tid; idx < 64 * D; idx += 256) {
        const int d = idx >> 6, r = idx & 63;
        CjT[idx] = (c0 + r < n) ? X[(c0 + r) * D + d] : 0.0;
      }
    if (tid < 64) vj[tid] = (c0 + tid < n) ? v[c0 + tid] : 0.0;
    __syncthreads();
    if (RADIAL1) {
      // one radial term: dimension loop outside, the thread's 16 columns inside; their coordinates
      // are wave-uniform and come from X through scalar loads (see gram_kernel); same operations in
      // the same order per pair
      const int cbase = __builtin_amdgcn_readfirstlane(ty) * 16;
      const double *xc = X + (c0 + cbase) * D;
      double s[16];
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) s[cc] = 0.0;
      for (int d = 0; d < D; ++d) {
        const double il = P.inv_len[0][d];
        const double r = RiT[d * 64 + tx];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
          const double u = (r - xc[cc * D + d]) * il;
          s[cc] += u * u;
        }
      }
      if (gi < n) {
        const int kind = P.kind[0];
        const double cs = P.c[0];
#pragma unroll 2
        for (int cc = 0; cc < 16; ++cc) {
          const int c = cbase + cc;
          const long gj = c0 + c;
          if (gj < n) {
            double f, dfdr2;
            radial_eval(kind, s[cc], f, dfdr2);
            double k = 0.0 + cs * f;
            if (gi == gj) k += P.noise_var;
            acc += k * vj[c];
          }
        }
      }
    } else if (gi < n) {
      const double *ri = RiT + tx;
      for (int cc = 0; cc < 16; ++cc) {
        const int c = ty * 16 + cc;
        const long gj = c0 + c;
        if (gj < n) {
          const double *cj = CjT + c;
          double k = simil_value(
              P, [&](int d) { return ri[d * 64]; }, [&](int d) { return cj[d * 64]; });
          if (gi == gj) k += P.noise_var;
          acc += k * vj[c];
        }
      }
    }
  }
  red[ty * 64 + tx] = acc;
  __syncthreads();
  if (ty == 0 && gi < npad)
    part[(long)slab * npad + gi] = red[tx] + red[64 + tx] + red[128 + tx] + red[192 + tx];
}

// r_i = y_i - sum_slab part[slab][i]  (i < n; 0 beyond); without y: the plain sum (K v over a range
// of columns, to be all-reduced)
__global__ void kmatvec_finish_kernel(const double *__restrict__ part, int nslab, long npad, long n,
                                      const double *__restrict__ y, double *__restrict__ r) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  double s = 0.0;
  for (int q = 0; q < nslab; ++q) s += part[(long)q * npad + i];
  r[i] = (i < n) ? (y ? y[i] - s : s) : 0.0;
}

void launch_residual(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                     int64_t npad, const double *v, const double *y, double *part, int nslab, double *r,
                     bool radial1) {
  const int ntile = (int)(npad / 64);
  const int tps = (ntile + nslab - 1) / nslab;
  const size_t lds = (size_t)(128 * ndim + 64 + 256) * sizeof(double);
  if (radial1)
    GOGP_KLAUNCH(kmatvec_kernel<true>, dim3((unsigned)ntile, (unsigned)nslab), dim3(256), lds, s, p, X, (long)n,
                       v, (long)npad, tps, part, 0, ntile);
  else
    GOGP_KLAUNCH(kmatvec_kernel<false>, dim3((unsigned)ntile, (unsigned)nslab), dim3(256), lds, s, p, X, (long)n,
                       v, (long)npad, tps, part, 0, ntile);
  GOGP_KLAUNCH(kmatvec_finish_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, part, nslab,
                     (long)npad, (long)n, y, r);
}

// out_i = sum over the columns j of share `part_idx` of `nparts` of K_ij v_j (exact fp64 K): the
// sharded form of the residual -- every rank takes its share of the column tiles, the shares are
// all-reduced.
void launch_kmatvec_share(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n, int64_t npad,
                          const double *v, int part_idx, int nparts, double *part, int nslab, double *out,
                          bool radial1) {
  const int ntile = (int)(npad / 64);
  const int per = (ntile + nparts - 1) / nparts;
  const int t0 = std::min(ntile, part_idx * per), t1 = std::min(ntile, (part_idx + 1) * per);
  const int tps = std::max(1, (t1 - t0 + nslab - 1) / nslab);
  const size_t lds = (size_t)(128 * ndim + 64 + 256) * sizeof(double);
  if (radial1)
    GOGP_KLAUNCH(kmatvec_kernel<true>, dim3((unsigned)ntile, (unsigned)nslab), dim3(256), lds, s, p, X, (long)n,
                       v, (long)npad, tps, part, t0, t1);
  else
    GOGP_KLAUNCH(kmatvec_kernel<false>, dim3((unsigned)ntile, (unsigned)nslab), dim3(256), lds, s, p, X, (long)n,
                       v, (long)npad, tps, part, t0, t1);
  GOGP_KLAUNCH(kmatvec_finish_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, part, nslab,
                     (long)npad, (long)n, (const double *)nullptr, out);
}

__global__ void prior_kernel(const DevParams *__restrict__ Pp, const double *__restrict__ Z,
                             long m, double *__restrict__ prior) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  const DevParams &P = *Pp;
  const double *z = Z + j * P.ndim;
  prior[j] = simil_value(
      P, [&](int d) { return z[d]; }, [&](int d) { return z[d]; });
}

template <class T>
static void gram_lower_t(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                         int64_t npad, T *K, int64_t ld) {
  const int nt = (int)(npad / 64);
  const int ntiles = nt * (nt + 1) / 2;
  const size_t lds = (size_t)2 * 64 * ndim * sizeof(double);
  GOGP_KLAUNCH((gram_kernel<false, T>), dim3(ntiles), dim3(256), lds, s, p, X, (long)n, X,
                     (long)n, K, (long)ld, nt, 0, 0, 0L);
}
void launch_gram_lower(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                       int64_t npad, double *K, int64_t ld) {
  gram_lower_t(s, p, ndim, X, n, npad, K, ld);
}

// The same lower triangle in two launches: block columns [0, wcols) on `s_first` (the panel
// chain can start on them while the rest is still being written), the remaining triangle on
// `s_rest`.  wcols is a multiple of 64.
template <class T>
static void gram_lower_split_t(hipStream_t s_first, hipStream_t s_rest, const DevParams *p, int ndim,
                               const double *X, int64_t n, int64_t npad, T *K, int64_t ld,
                               int64_t wcols) {
  const int nt = (int)(npad / 64);
  int w = (int)(wcols / 64);
  if (w > nt) w = nt;
  const size_t lds = (size_t)2 * 64 * ndim * sizeof(double);
  const unsigned nz = (unsigned)tl_batch.k;
  GOGP_KLAUNCH((gram_kernel<false, T>), dim3(nt * w, 1, nz), dim3(256), lds, s_first, p, X, (long)n, X,
                     (long)n, K, (long)ld, nt, w, 0, tl_batch.stride);
  const int nr = nt - w;
  if (nr > 0)
    GOGP_KLAUNCH((gram_kernel<false, T>), dim3(nr * (nr + 1) / 2, 1, nz), dim3(256), lds, s_rest, p, X,
                       (long)n, X, (long)n, K, (long)ld, nt, 0, w, tl_batch.stride);
}
void launch_gram_lower_split(hipStream_t s_first, hipStream_t s_rest, const DevParams *p, int ndim,
                             const double *X, int64_t n, int64_t npad, double *K, int64_t ld,
                             int64_t wcols) {
  gram_lower_split_t(s_first, s_rest, p, ndim, X, n, npad, K, ld, wcols);
}
void launch_gram_lower_split(hipStream_t s_first, hipStream_t s_rest, const DevParams *p, int ndim,
                             const double *X, int64_t n, int64_t npad, float *K, int64_t ld,
                             int64_t wcols) {
  gram_lower_split_t(s_first, s_rest, p, ndim, X, n, npad, K, ld, wcols);
}

template <class T>
static void gram_local_t(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                         int64_t mrows, int64_t ncols, BlockMap map, T *K, int64_t ld) {
  const int ntr = (int)(mrows / 64), ntc = (int)(ncols / 64);
  if (ntr <= 0 || ntc <= 0) return;
  const size_t lds = (size_t)2 * 64 * ndim * sizeof(double);
  GOGP_KLAUNCH(gram_local_kernel<T>, dim3(ntr * ntc), dim3(256), lds, s, p, X, (long)n, K, (long)ld,
                     ntc, map);
}
void launch_gram_local(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                       int64_t mrows, int64_t ncols, BlockMap map, double *K, int64_t ld) {
  gram_local_t(s, p, ndim, X, n, mrows, ncols, map, K, ld);
}
void launch_gram_local(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                       int64_t mrows, int64_t ncols, BlockMap map, float *K, int64_t ld) {
  gram_local_t(s, p, ndim, X, n, mrows, ncols, map, K, ld);
}

template <class T>
static void cross_t(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                    int64_t npad, const double *Z, int64_t m, int64_t mpad, T *KsT, int64_t ld) {
  const int ntr = (int)(mpad / 64), ntc = (int)(npad / 64);
  const size_t lds = (size_t)2 * 64 * ndim * sizeof(double);
  GOGP_KLAUNCH((gram_kernel<true, T>), dim3(ntr * ntc), dim3(256), lds, s, p, Z, (long)m, X,
                     (long)n, KsT, (long)ld, ntc, 0, 0, 0L);
}
void launch_cross(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                  int64_t npad, const double *Z, int64_t m, int64_t mpad, double *KsT, int64_t ld) {
  cross_t(s, p, ndim, X, n, npad, Z, m, mpad, KsT, ld);
}
void launch_cross(hipStream_t s, const DevParams *p, int ndim, const double *X, int64_t n,
                  int64_t npad, const double *Z, int64_t m, int64_t mpad, float *KsT, int64_t ld) {
  cross_t(s, p, ndim, X, n, npad, Z, m, mpad, KsT, ld);
}

void launch_prior(hipStream_t s, const DevParams *p, const double *Z, int64_t m,
                  double *prior) {
  if (m <= 0) return;
  GOGP_KLAUNCH(prior_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, p, Z,
                     (long)m, prior);
}

}  // namespace gogp
