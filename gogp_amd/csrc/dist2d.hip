// dist2d.hip -- ONE evaluation sharded over several GPUs: 2-D block-cyclic layout.
//
// Reference counterpart: none.  gp.GP.absorb / Gradient run in one process
// (goroutines only, gp/gp.go:165-213, :436-470); this file distributes the same
// mathematics (Gram build gp/gp.go:109-156, Cholesky :228, alpha :232-236, LML :244-253,
// gradient :418-499) over a Pr x Pc grid of GPUs, one process per GPU.
//
// Layout.  nb = 512 (two 256-wide panels of the single-GPU sweep).  The padded matrix has
// NB = npad / nb block rows / columns, npad a multiple of nb * Pc (Pr divides Pc).  Tile (I, J)
// lives on the rank at grid position (I mod Pr, J mod Pc); rank = pr * Pc + pc.  Per rank:
//   A    (mloc*nb) x (nloc*nb) row-major, mloc = NB/Pr, nloc = NB/Pc: its tiles of K.  Tiles of
//        the global lower triangle: Gram matrix -> destroyed by the trailing updates -> finally
//        K^-1.  Tiles of strictly upper blocks: R, the right-hand side of the triangular inverse.
//   Lch  nloc chunks of (mloc*nb) x nb (leading dimension nb): chunk bj = this rank's tiles of
//        block column bj*Pc + pc of L.  A chunk is contiguous, so the tiles a process row needs
//        travel without packing.
//   Ych  the same for Y = L^-T (upper triangular: chunk P holds row blocks <= P).
//   Dinv NB inverses (nb x nb) of the diagonal tiles of L, replicated.
//   four pairs of panel buffers (two steps in flight): the tiles of the current L / Y panel
//        this rank needs for its tile ROWS (Lrow, Yrow: mloc*nb x nb) and for its tile COLUMNS
//        (Lcol, Ycol: nloc*nb x nb).
//
// Step P (block column P; kr = P mod Pr, kc = P mod Pc):
//   1. rank (kr, kc) factors the diagonal tile (two 256-blocks on the one-workgroup kernel +
//      three small products) and inverts it; the inverse goes to every rank            [D_P]
//   2. process column kc: L[I, P] = A[I, P] inv(L_PP)^T for its tile rows I > P  (one GEMM)
//   3. exchange [L_P]: every rank of process column kc sends its chunk to the ranks of its
//      process ROW (they share its tile rows), and the tiles J = pc' (mod Pc) of it to the
//      ranks (pr' != pr, pc') whose tile COLUMNS they are.  Because Pr divides Pc all tiles a
//      rank needs for its columns come from ONE sender, (pc mod Pr, kc); ranks with
//      pc mod Pr == pr find them inside their row chunk.  Direct peer sends, no ring: on the
//      fully connected xGMI mesh every pair has its own link.
//   4. trailing update A[I, J] -= L[I, P] L[J, P]^T of the local tiles with I >= J > P on
//      MFMA (dgemm.hip, tile filter rule 1); the next panel's block column first, on the
//      chain stream (look-ahead), the rest on the bulk stream.
//   5. process column kc: Y[I, P] = R[I, P] inv(L_PP)^T (I < P), Y[P, P] = inv(L_PP)^T;
//      z_P += Y[I, P]^T y_I (partial sums of z = L^-1 y = Y^T y)
//   6. exchange [Y_P], same pattern as 3 with the tile rows / columns <= P
//   7. R[I, J] -= Y[I, P] L[J, P]^T (I <= P < J) and, for the gradient,
//      K^-1[I, J] (+)= Y[I, P] Y[J, P]^T (J <= I <= P; tile filter rule 2)
// End: one all-reduce gives z, the log-determinant and the failure flags; alpha = Y z from the
// local chunks and a second all-reduce; LML on the host.  Gradient: fused reduction over the
// local tiles of K^-1 (grad.hip) and one all-reduce of the slot sums.
//
// Streams: sp chain (diagonal tile, panel solve, look-ahead update), s bulk trailing update,
// st chain of the inverse, s2 its bulk updates, sc communication.  With the RCCL transport
// nothing synchronises with the host inside the sweep.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "comm.h"
#include "handle.h"

using namespace gogp;

namespace gogp {

// The triangular inverse runs INV_LAG block columns behind the Cholesky sweep in the enqueue
// order (so that the Cholesky chain never queues behind an exchange of the inverse on the
// single, in-order communication stream); the L panel buffers therefore live INV_LAG steps
// longer: a ring of LRING = INV_LAG + 2.
constexpr int INV_LAG = 2;
constexpr int LRING = INV_LAG + 2;

struct Dist2D {
  int rank = 0, nranks = 1, Pr = 1, Pc = 1, pr = 0, pc = 0;
  int nb = 512, tpb = 4, tpb_shift = 2, nb_shift = 9;
  int NB = 0, mloc = 0, nloc = 0;
  Transport *tr = nullptr;
  hipStream_t sc = nullptr;
  double *A = nullptr, *Lch = nullptr, *Ych = nullptr, *Dinv = nullptr;
  double *Lrow[LRING] = {nullptr, nullptr, nullptr, nullptr}, *Lcol[LRING] = {nullptr, nullptr, nullptr, nullptr};
  double *Yrow[2] = {nullptr, nullptr}, *Ycol[2] = {nullptr, nullptr};
  double *pack = nullptr;   // send staging of the strided column pieces
  double *scr = nullptr;    // 2 x 256 x 256 scratch of the diagonal tile
  double *yloc = nullptr;   // y at the local rows
  double *rloc = nullptr;   // fp32 evaluation: the residual at the local rows
  double *red = nullptr;    // [z (npad) | logdet | failing pivot + 1 of every rank]
  double *ared = nullptr;   // alpha partial sums (npad)
  double *gpart = nullptr;
  double *tpart = nullptr;  // scratch of the z partial sums (chunk_tdot)
  int64_t cap_npad = 0;
  int64_t bytes = 0;
  double *t64 = nullptr;    // fp32 evaluation: fp64 images of one diagonal tile (A, L, inverse: 3 x nb x nb)
  // fp32 evaluation, option "diag_fp64" (diagsyrk.hip): this rank's tiles of the GLOBAL diagonal kept in fp64 -- every
  // panel's contribution summed in fp64 from the float panel tiles.  They are the local tile rows dq_first, dq_first +
  // dq_step, ... (dq_count of them; none when the rank's grid row and column never meet on the diagonal).
  double *d64 = nullptr;
  int dq_first = 0, dq_step = 1, dq_count = 0;
  bool d64_on = false;      // set by the evaluation in progress
  int ldA() const { return nloc * nb; }
  // The matrix buffers above (A, Lch, Ych, Dinv, the panel rings, pack) hold T = double or -- on a
  // handle with precision 32 -- float elements; typed views:
  template <class T> T *mat(double *p) const { return reinterpret_cast<T *>(p); }
  template <class T> T *lchunk(int bj) const { return reinterpret_cast<T *>(Lch) + (size_t)bj * mloc * nb * nb; }
  template <class T> T *ychunk(int bj) const { return reinterpret_cast<T *>(Ych) + (size_t)bj * mloc * nb * nb; }
  int rank_of(int r, int c) const { return r * Pc + c; }
  BlockMap map() const {
    BlockMap m;
    m.nb_shift = nb_shift;
    m.pr = pr;
    m.Pr = Pr;
    m.pc = pc;
    m.Pc = Pc;
    return m;
  }
};

}  // namespace gogp

// a transfer of `elems` elements of type T through the transport's double-typed interface (tile
// sizes are even, so float payloads travel as half as many doubles: the transports only move bytes)
template <class T>
static inline XferOp xop(int peer, bool send, T *p, int64_t elems) {
  return XferOp{peer, send, reinterpret_cast<double *>(p), elems * (int64_t)sizeof(T) / (int64_t)sizeof(double)};
}
template <class T>
static inline void pack_blocks_t(hipStream_t s, T *dst, const T *src, int nblk, int64_t blk_elems, int first,
                                 int stride) {
  launch_pack_blocks(s, reinterpret_cast<double *>(dst), reinterpret_cast<const double *>(src), nblk,
                     blk_elems * (int64_t)sizeof(T) / (int64_t)sizeof(double), first, stride);
}

// number of local blocks b (global index b*Pn + p) with global index <= P
static inline int first_gt(int P, int p, int Pn) { return P >= p ? (P - p) / Pn + 1 : 0; }

enum { EDIAG = 0, ED = 1, EPANEL = 2, EL = 3, EUPD = 4, ELA = 5, EYCH = 6, EY = 7, ERUPD = 8, ERLA = 9,
       ENEV = 10 };
static inline size_t E(int P, int k) { return EV_BASE + (size_t)ENEV * P + k; }
static inline void rec(gogp_handle *h, size_t i, hipStream_t s) { (void)hipEventRecord(ev(h, i), s); }
static inline void wait(gogp_handle *h, hipStream_t s, size_t i) { (void)hipStreamWaitEvent(s, ev(h, i), 0); }

static void dist_free_n(Dist2D *d) {
  for (double *p : {d->A, d->Lch, d->Ych, d->Dinv, d->Yrow[0], d->Yrow[1], d->Ycol[0], d->Ycol[1], d->pack,
                    d->yloc, d->rloc, d->red, d->ared, d->gpart, d->tpart, d->d64})
    (void)hipFree(p);
  d->rloc = nullptr;
  for (int i = 0; i < LRING; ++i) {
    (void)hipFree(d->Lrow[i]);
    (void)hipFree(d->Lcol[i]);
    d->Lrow[i] = d->Lcol[i] = nullptr;
  }
  d->A = d->Lch = d->Ych = d->Dinv = d->pack = d->yloc = d->red = d->ared = d->gpart = d->tpart = d->d64 = nullptr;
  for (int i = 0; i < 2; ++i) d->Yrow[i] = d->Ycol[i] = nullptr;
  d->cap_npad = 0;
  d->bytes = 0;
}

void gogp_dist_destroy(gogp_handle *h) {
  Dist2D *d = h->dist;
  if (!d) return;
  if (d->sc) (void)hipStreamSynchronize(d->sc);
  dist_free_n(d);
  (void)hipFree(d->scr);
  (void)hipFree(d->t64);
  delete d->tr;
  if (d->sc) (void)hipStreamDestroy(d->sc);
  delete d;
  h->dist = nullptr;
}

int gogp_dist_sync(gogp_handle *h) {
  if (h->dist && h->dist->sc) HIPCHK(h, hipStreamSynchronize(h->dist->sc));
  return GOGP_OK;
}

extern "C" int gogp_dist_grid(int nranks, int *prow, int *pcol) {
  if (nranks < 1 || !prow || !pcol) return GOGP_EARG;
  int best = 1;
  for (int r = 1; r * r <= nranks; ++r)
    if (nranks % r == 0 && (nranks / r) % r == 0) best = r;
  *prow = best;
  *pcol = nranks / best;
  return GOGP_OK;
}

extern "C" int gogp_dist_unique_id(void *id128) {
  if (!id128) return GOGP_EARG;
  return rccl_unique_id(id128);
}

extern "C" int64_t gogp_dist_local_bytes(const gogp_handle *h) {
  return (h && h->dist) ? h->dist->bytes : 0;
}

static int dist_init_common(gogp_handle *h, int rank, int nranks, int prow, int pcol, Transport *tr) {
  if (prow < 1 || pcol < 1 || prow * pcol != nranks || pcol % prow != 0 || rank < 0 || rank >= nranks ||
      nranks > 64) {
    delete tr;
    return fail(h, GOGP_EARG, "dist_init: the grid must be Pr x Pc = nranks with Pr dividing Pc");
  }
  if (h->dist) gogp_dist_destroy(h);
  Dist2D *d = new Dist2D();
  d->rank = rank;
  d->nranks = nranks;
  d->Pr = prow;
  d->Pc = pcol;
  d->pr = rank / pcol;
  d->pc = rank % pcol;
  d->tr = tr;
  // (normal priority: with a high-priority communication stream four gloo ranks sharing one GPU ran
  // an evaluation at N = 8192 in 771 ms instead of 84 ms)
  hipError_t e = hipStreamCreateWithFlags(&d->sc, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&d->scr, (size_t)2 * PANEL * PANEL * sizeof(double));
  if (e == hipSuccess && h->prec == 32) e = hipMalloc(&d->t64, (size_t)3 * d->nb * d->nb * sizeof(double));
  h->dist = d;
  if (e != hipSuccess) {
    gogp_dist_destroy(h);
    return fail(h, GOGP_EHIP, "dist_init: HIP allocation failed");
  }
  // data loaded before the handle was sharded must be loaded again (other buffers)
  h->have_data = h->factored = h->observed = h->grad_valid = h->have_kinv = false;
  return GOGP_OK;
}

extern "C" int gogp_dist_init_rccl(gogp_handle *h, int rank, int nranks, int prow, int pcol,
                                   const void *id128) {
  if (!h || !id128) return GOGP_EARG;
  if (hipSetDevice(h->device) != hipSuccess) return fail(h, GOGP_EHIP, "hipSetDevice failed");
  std::string err;
  Transport *tr = make_rccl_transport(rank, nranks, id128, &err);
  if (!tr) {
    h->err = err;
    return GOGP_EHIP;
  }
  return dist_init_common(h, rank, nranks, prow, pcol, tr);
}

// measurement only: rank `rank` of a prow x pcol grid alone on this GPU behind the replay transport (comm.h).  A C++
// symbol for libgogp_testhooks.so (gogp_test_dist_init_replay), not part of the C ABI.
namespace gogp {
int dist_init_replay(gogp_handle *h, int rank, int nranks, int prow, int pcol) {
  if (!h) return GOGP_EARG;
  return dist_init_common(h, rank, nranks, prow, pcol, make_replay_transport(rank, nranks));
}
}  // namespace gogp

extern "C" int gogp_dist_init_callbacks(gogp_handle *h, int rank, int nranks, int prow, int pcol,
                                        gogp_exchange_fn exchange, gogp_allreduce_fn allreduce,
                                        void *user) {
  if (!h || !exchange || !allreduce) return GOGP_EARG;
  if (hipSetDevice(h->device) != hipSuccess) return fail(h, GOGP_EHIP, "hipSetDevice failed");
  return dist_init_common(h, rank, nranks, prow, pcol,
                          make_callback_transport(rank, nranks, exchange, allreduce, user));
}

// ---- pre-flight of the transport (bench.py, tests): the two primitives of the sweep, alone ------------
extern "C" int gogp_dist_comm_ranks(const gogp_handle *h, int *is_rccl) {
  if (!h || !h->dist || !h->dist->tr) return -1;
  if (is_rccl) *is_rccl = h->dist->tr->is_rccl() ? 1 : 0;
  return h->dist->tr->comm_ranks();
}

__global__ void selftest_fill_kernel(double *p, long count, double base) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) p[i] = base + (double)(i & 1023);
}

// phase 0: ONE group holding a send to rank+1 and a receive from rank-1 (the shape of every panel
// exchange of the sweep: grouped ncclSend / ncclRecv between peers), payload checked on the host;
// phase 1: one all-reduce, sum checked.  Synchronises the communication stream: the caller's
// watchdog sees a hang as a call that does not return.
extern "C" int gogp_dist_selftest(gogp_handle *h, int phase, int64_t count) {
  if (!h || !h->dist || !h->dist->tr) return fail(h, GOGP_ESTATE, "selftest: not a sharded handle");
  if (count < 1 || count > (1 << 24) || phase < 0 || phase > 1) return fail(h, GOGP_EARG, "selftest: bad arguments");
  Dist2D *d = h->dist;
  HIPCHK(h, hipSetDevice(h->device));
  const int n = d->nranks, r = d->rank;
  double *buf = nullptr;
  HIPCHK(h, hipMalloc(&buf, (size_t)2 * count * sizeof(double)));
  std::vector<double> host((size_t)count);
  const unsigned nblk = (unsigned)((count + 255) / 256);
  int rc = GOGP_OK;
  std::string terr;
  hipError_t e = hipSuccess;
  char msg[200] = "";
  if (phase == 0) {
    hipLaunchKernelGGL(selftest_fill_kernel, dim3(nblk), dim3(256), 0, d->sc, buf, (long)count, 1000.0 * (r + 1));
    e = hipMemsetAsync(buf + count, 0, (size_t)count * sizeof(double), d->sc);
    if (n > 1 && e == hipSuccess) {
      std::vector<XferOp> ops;
      ops.push_back(XferOp{(r + 1) % n, true, buf, count});
      ops.push_back(XferOp{(r + n - 1) % n, false, buf + count, count});
      rc = d->tr->group(d->sc, ops, &terr);
    }
    if (rc == GOGP_OK && e == hipSuccess)
      e = hipMemcpyAsync(host.data(), n > 1 ? buf + count : buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, d->sc);
    if (rc == GOGP_OK && e == hipSuccess) e = hipStreamSynchronize(d->sc);
    if (rc == GOGP_OK && e == hipSuccess) {
      const double base = 1000.0 * (((r + n - 1) % n) + 1);
      for (int64_t i = 0; i < count; ++i)
        if (host[(size_t)i] != base + (double)(i & 1023)) {
          snprintf(msg, sizeof msg, "selftest: send/recv ring delivered a wrong payload at element %lld (rank %d)",
                   (long long)i, r);
          rc = GOGP_EHIP;
          break;
        }
    }
  } else {
    hipLaunchKernelGGL(selftest_fill_kernel, dim3(nblk), dim3(256), 0, d->sc, buf, (long)count, (double)(r + 1));
    rc = d->tr->allreduce(d->sc, buf, count, &terr);
    if (rc == GOGP_OK) e = hipMemcpyAsync(host.data(), buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, d->sc);
    if (rc == GOGP_OK && e == hipSuccess) e = hipStreamSynchronize(d->sc);
    if (rc == GOGP_OK && e == hipSuccess) {
      const double base = 0.5 * n * (n + 1);
      for (int64_t i = 0; i < count; ++i)
        if (host[(size_t)i] != base + (double)n * (double)(i & 1023)) {
          snprintf(msg, sizeof msg, "selftest: all-reduce returned a wrong sum at element %lld (rank %d)", (long long)i, r);
          rc = GOGP_EHIP;
          break;
        }
    }
  }
  (void)hipFree(buf);
  if (rc != GOGP_OK) {
    h->err = msg[0] ? std::string(msg) : "selftest: " + terr;
    return rc;
  }
  HIPCHK(h, e);
  return GOGP_OK;
}

// ---- data: sizes and buffers of this rank's shard ----------------------------------------------
#define DMALLOC(ptr, count)                                                     \
  do {                                                                          \
    const size_t b_ = (size_t)(count) * sizeof(double);                         \
    HIPCHK(h, hipMalloc(&(ptr), b_ ? b_ : sizeof(double)));                     \
    d->bytes += (int64_t)b_;                                                    \
  } while (0)

int gogp_dist_ensure_n(gogp_handle *h, int64_t n) {
  Dist2D *d = h->dist;
  const int64_t unit = (int64_t)d->nb * d->Pc;
  const int64_t npad = n <= 0 ? 0 : ((n + unit - 1) / unit) * unit;
  h->n = n;
  h->npad = npad;
  h->nblk = (int)(npad / TILE);
  h->factored = h->have_alpha = h->have_kinv = h->observed = h->grad_valid = false;
  h->trtri_done = false;
  d->NB = (int)(npad / d->nb);
  d->mloc = d->NB / d->Pr;
  d->nloc = d->NB / d->Pc;
  {
    // this rank's tiles of the global diagonal: local tile row bi is global I = pr + Pr bi, mine iff I = pc (mod Pc);
    // Pr | Pc, so: iff pc = pr (mod Pr) and bi = (pc - pr) / Pr (mod Pc / Pr)
    const int qq = d->Pc / d->Pr;
    d->dq_step = qq;
    d->dq_first = 0;
    d->dq_count = 0;
    if ((d->pc - d->pr) % d->Pr == 0) {
      d->dq_first = (((d->pc - d->pr) / d->Pr) % qq + qq) % qq;
      if (d->mloc > d->dq_first) d->dq_count = (d->mloc - d->dq_first + qq - 1) / qq;
    }
  }
  if (npad > d->cap_npad) {
    // the unsharded N x N buffers are never allocated on a sharded handle
    for (double **p : {&h->dX, &h->dy, &h->bufA, &h->bufL, &h->bufY, &h->Dinv, &h->z, &h->w, &h->alpha,
                       &h->gpart, &h->rw, &h->rz, &h->rd, &h->rpart}) {
      (void)hipFree(*p);
      *p = nullptr;
    }
    h->cap_npad = 0;
    h->cap_y = 0;
    dist_free_n(d);
    const size_t nb2 = (size_t)d->nb * d->nb;
    const size_t mrows = (size_t)d->mloc * d->nb, ncols = (size_t)d->nloc * d->nb;
    // (+ GOGP_MAX_NDIM doubles of slack: grad.hip reads a few coordinates past the last row)
    HIPCHK(h, hipMalloc(&h->dX, ((size_t)npad * h->D + GOGP_MAX_NDIM) * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->dy, (size_t)npad * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->z, (size_t)npad * sizeof(double)));
    HIPCHK(h, hipMalloc(&h->alpha, (size_t)npad * sizeof(double)));
    d->bytes = (int64_t)((size_t)npad * (h->D + 3) * sizeof(double));
    // matrix buffers: float elements on a precision-32 handle (counts in doubles, rounded up)
    const size_t es = h->esz();
    auto dbl = [&](size_t elems) { return (elems * es + sizeof(double) - 1) / sizeof(double); };
    DMALLOC(d->A, dbl(mrows * ncols));
    DMALLOC(d->Lch, dbl(mrows * ncols));
    DMALLOC(d->Ych, dbl(mrows * ncols));
    DMALLOC(d->Dinv, dbl((size_t)d->NB * nb2));
    for (int i = 0; i < LRING; ++i) {
      DMALLOC(d->Lrow[i], dbl(mrows * d->nb));
      DMALLOC(d->Lcol[i], dbl(ncols * d->nb));
    }
    for (int i = 0; i < 2; ++i) {
      DMALLOC(d->Yrow[i], dbl(mrows * d->nb));
      DMALLOC(d->Ycol[i], dbl(ncols * d->nb));
    }
    DMALLOC(d->pack, dbl((size_t)(d->Pc / d->Pr) * ncols * d->nb));
    DMALLOC(d->yloc, mrows);
    if (h->prec == 32) {  // refinement of alpha: residual scratch
      DMALLOC(d->rloc, mrows);
      HIPCHK(h, hipMalloc(&h->rw, (size_t)npad * sizeof(double)));
      HIPCHK(h, hipMalloc(&h->rpart, (size_t)REFINE_SLABS * npad * sizeof(double)));
      d->bytes += (int64_t)((size_t)(1 + REFINE_SLABS) * npad * sizeof(double));
    }
    if (h->prec == 32 && d->dq_count > 0) DMALLOC(d->d64, (size_t)d->dq_count * nb2);  // (npad is the largest so far: so is dq_count)
    DMALLOC(d->red, (size_t)npad + 1 + d->nranks);
    DMALLOC(d->ared, (size_t)npad);
    DMALLOC(d->gpart, (size_t)grad_reduce_blocks_local((int64_t)mrows, (int64_t)ncols) * NACC);
    DMALLOC(d->tpart, (size_t)chunk_tdot_scratch((int64_t)mrows, d->nb));
    d->cap_npad = npad;
  }
  return GOGP_OK;
}

__global__ void residual_from_kernel(double *__restrict__ kv, const double *__restrict__ y, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) kv[i] = y[i] - kv[i];
}
static void launch_residual_from(hipStream_t s, double *kv, const double *y, int64_t count) {  // kv := y - kv
  if (count <= 0) return;
  hipLaunchKernelGGL(residual_from_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, kv, y, (long)count);
}

__global__ void gather_rows_kernel(const double *__restrict__ y, double *__restrict__ yloc, long rows,
                                   BlockMap map) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) yloc[i] = y[map.grow(i)];
}

// ---- the diagonal tile: factor + inverse of one nb x nb block on the chain stream -----------
// Always fp64 (as the 256-blocks of the single-GPU path): Ablk (lda) in, Lblk (ld nb) and its dense
// inverse Dv (ld nb) out; the log-determinant is accumulated from the fp64 factor.
// Dv (ld nb) holds X00 and X11, the inverses of the two diagonal 256-blocks of a 512x512 lower
// triangular tile whose lower-left block is L10 (ld nb): fill in X10 = -X11 L10 X00 and the zero block
static void tile_inverse_offdiag(Dist2D *d, hipStream_t sp, const double *L10, double *Dv, GemmProfile *pf) {
  const int nb = d->nb;
  launch_zero_block(sp, Dv + PANEL, nb, PANEL, PANEL);
  double *X00T = d->scr, *WT = d->scr + (size_t)PANEL * PANEL;
  launch_transpose_sq(sp, Dv, nb, X00T, PANEL, PANEL);
  // WT = (L10 X00)^T = X00^T L10^T
  launch_dgemm_nt(sp, GEMM_RECT, 2, 2, PANEL, 1.0, X00T, PANEL, L10, nb, 0.0, WT, PANEL, pf);
  // X10 = -X11 (L10 X00) = -X11 WT^T
  launch_dgemm_nt(sp, GEMM_RECT, 2, 2, PANEL, -1.0, Dv + (size_t)PANEL * nb + PANEL, nb, WT, PANEL, 0.0,
                  Dv + (size_t)PANEL * nb, nb, pf);
}

static void diag_tile64(gogp_handle *h, Dist2D *d, hipStream_t sp, double *Ablk, int lda, double *Lblk,
                        double *Dv, int64_t row0) {
  const int nb = d->nb;
  GemmProfile *pf = &h->prof;
  // L_PP = [[L00, 0], [L10, L11]];  inv = [[X00, 0], [-X11 L10 X00, X11]]
  launch_diag256_ld512(sp, Ablk, lda, Lblk, nb, Dv, row0, h->n, h->info);
  double *L10 = Lblk + (size_t)PANEL * nb;
  launch_dgemm_nt(sp, GEMM_RECT, 2, 2, PANEL, 1.0, Ablk + (size_t)PANEL * lda, lda, Dv, nb, 0.0, L10, nb,
                  pf);
  launch_dgemm_nt(sp, GEMM_LOWER, 2, 2, PANEL, -1.0, L10, nb, L10, nb, 1.0,
                  Ablk + (size_t)PANEL * lda + PANEL, lda, pf);
  launch_diag256_ld512(sp, Ablk + (size_t)PANEL * lda + PANEL, lda, Lblk + (size_t)PANEL * nb + PANEL, nb,
                       Dv + (size_t)PANEL * nb + PANEL, row0 + PANEL, h->n, h->info);
  launch_zero_block(sp, Lblk + PANEL, nb, PANEL, PANEL);
  tile_inverse_offdiag(d, sp, L10, Dv, pf);
  launch_logdet_block(sp, Lblk, nb, row0, h->n, nb, d->red + h->npad);
}
static void diag_tile(gogp_handle *h, Dist2D *d, hipStream_t sp, int P, int bi_d, int bj_d, double) {
  const int nb = d->nb, ldA = d->ldA();
  const size_t nb2 = (size_t)nb * nb;
  diag_tile64(h, d, sp, d->A + (size_t)bi_d * nb * ldA + (size_t)bj_d * nb, ldA,
              d->lchunk<double>(bj_d) + (size_t)bi_d * nb2, d->Dinv + (size_t)P * nb2, (int64_t)P * nb);
}
// fp32 evaluation: the tile is widened into fp64 scratch, factored and inverted there, and the
// factor and the inverse are rounded to float once (api.hip: diag_block does the same per 256-block)
static void diag_tile(gogp_handle *h, Dist2D *d, hipStream_t sp, int P, int bi_d, int bj_d, float) {
  const int nb = d->nb, ldA = d->ldA();
  const size_t nb2 = (size_t)nb * nb;
  float *Ablk = d->mat<float>(d->A) + (size_t)bi_d * nb * ldA + (size_t)bj_d * nb;
  float *Lblk = d->lchunk<float>(bj_d) + (size_t)bi_d * nb2;
  float *Dv = d->mat<float>(d->Dinv) + (size_t)P * nb2;
  double *A64 = d->t64, *L64 = d->t64 + nb2, *D64 = d->t64 + 2 * nb2;
  if (d->d64_on)  // option "diag_fp64": the tile as its fp64 image accumulated it (diagsyrk.hip), not the float one
    A64 = d->d64 + (size_t)((bi_d - d->dq_first) / d->dq_step) * nb2;
  else
    launch_convert_block(sp, Ablk, ldA, A64, nb, nb, nb);
  diag_tile64(h, d, sp, A64, nb, L64, D64, (int64_t)P * nb);
  launch_convert_block(sp, L64, nb, Lblk, nb, nb, nb);
  launch_convert_block(sp, D64, nb, Dv, nb, nb, nb);
}

#define TRCHK(call)                            \
  do {                                         \
    std::string e_;                            \
    int r_ = (call);                           \
    if (r_ != GOGP_OK) {                       \
      h->err = "sharded evaluation: " + e_;    \
      return r_;                               \
    }                                          \
  } while (0)

static void widen_tile(hipStream_t s, const float *src, int lds_, double *dst, int nb) {
  launch_convert_block(s, src, lds_, dst, nb, nb, nb);
}
static void widen_tile(hipStream_t, const double *, int, double *, int) {}
// my fp64 diagonal tiles t0 .. t0 + cnt - 1 -= their panel tiles (local tile rows dq_first + t dq_step of Lrow) times themselves
static void d64_tiles_update(Dist2D *d, hipStream_t s, const float *Lrow, int t0, int cnt) {
  const size_t nb2 = (size_t)d->nb * d->nb;
  launch_diag_syrk_f64_tiles(s, Lrow + (size_t)(d->dq_first + t0 * d->dq_step) * nb2, d->nb, d->nb, d->d64 + (size_t)t0 * nb2,
                             cnt, (int64_t)d->dq_step * (int64_t)nb2, d->nb);
}
static void d64_tiles_update(Dist2D *, hipStream_t, const double *, int, int) {}

// ---- one sharded evaluation: Gram + Cholesky + triangular inverse (+ K^-1) -------------------
template <class T>
static int dist_factorize_t(gogp_handle *h, bool want_kinv) {
  Dist2D *d = h->dist;
  T *const A = d->mat<T>(d->A), *const Dinv = d->mat<T>(d->Dinv), *const packb = d->mat<T>(d->pack);
  const int nb = d->nb, tpb = d->tpb, Pr = d->Pr, Pc = d->Pc, pr = d->pr, pc = d->pc;
  const int NB = d->NB, mloc = d->mloc, nloc = d->nloc, ldA = d->ldA();
  const int q = Pc / Pr;
  const size_t nb2 = (size_t)nb * nb;
  const int64_t npad = h->npad;
  hipStream_t s = h->s, sp = h->sp, st = h->st, s2 = h->s2, sc = d->sc;
  GemmProfile *pf = &h->prof;
  // evaluations are host-synchronous at their boundaries on a sharded handle
  for (hipStream_t qs : {s, sp, st, s2, h->sl, sc}) HIPCHK(h, hipStreamSynchronize(qs));
  h->trtri_pending = h->alpha_pending = false;
  h->factored = h->have_alpha = h->have_kinv = h->grad_valid = false;
  h->trtri_done = false;
  h->notpd = -1;
  int rc = gogp_upload_params(h);  // on h->s
  if (rc != GOGP_OK) return rc;
  HIPCHK(h, hipMemsetAsync(h->info, 0, sizeof(long long), s));
  HIPCHK(h, hipMemsetAsync(d->red, 0, ((size_t)npad + 1 + d->nranks) * sizeof(double), s));
  HIPCHK(h, hipMemsetAsync(d->ared, 0, (size_t)npad * sizeof(double), s));
  const long lrows = (long)mloc * nb;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((lrows + 255) / 256)), dim3(256), 0, s, h->dy,
                     d->yloc, lrows, d->map());
  // every rank builds its own tiles of K (X is replicated): no communication for the O(N^2) step;
  // the strictly upper blocks are zero-filled (R)
  launch_gram_local(s, h->devP, h->D, h->dX, h->n, (int64_t)mloc * nb, (int64_t)nloc * nb, d->map(), A,
                    ldA);
  // fp32 evaluation, option "diag_fp64": my tiles of the global diagonal leave the float matrix here (widened once; every
  // later contribution is summed in fp64 from the float panel tiles: diagsyrk.hip, api.hip does the same per 256-block)
  d->d64_on = sizeof(T) == 4 && h->diag_fp64 != 0 && d->dq_count > 0 && d->d64 != nullptr;
  if (d->d64_on)
    for (int t = 0; t < d->dq_count; ++t) {
      const int bi = d->dq_first + t * d->dq_step, bj = (pr + Pr * bi - pc) / Pc;
      widen_tile(s, A + (size_t)bi * nb * ldA + (size_t)bj * nb, ldA, d->d64 + (size_t)t * nb2, nb);
    }
  rec(h, EV_GRAM, s);
  for (hipStream_t qs : {sp, st, s2, sc}) wait(h, qs, EV_GRAM);

  std::vector<XferOp> ops;
  // ---- Cholesky part of block column P (sections 1-4) --------------------------------------------
  auto chol_step = [&](int P) -> int {
    const int kr = P % Pr, kc = P % Pc;
    const int bi_d = P / Pr, bj_d = P / Pc;
    const int bi0 = first_gt(P, pr, Pr);  // local row blocks [0, bi0): global <= P; [bi0, mloc): > P
    const int bj0 = first_gt(P, pc, Pc);
    const bool in_col = (pc == kc), in_row = (pr == kr), is_diag = in_col && in_row;
    const int lslot = P % LRING;
    const int diag_rank = d->rank_of(kr, kc);
    T *Dv = Dinv + (size_t)P * nb2;
    const bool next_mine = (P + 1 < NB) && (pc == (P + 1) % Pc);  // I hold tiles of block column P+1

    // ---- 1. diagonal tile -----------------------------------------------------------------
    if (in_col && P >= 2) wait(h, sp, E(P - 2, EUPD));  // bulk update of step P-2 touched column P
    if (is_diag) {
      diag_tile(h, d, sp, P, bi_d, bj_d, T());
      rec(h, E(P, EDIAG), sp);
      wait(h, sc, E(P, EDIAG));
    }
    // The inverse gates the panel solve and the Y panel of process COLUMN kc only: this group carries it
    // to those Pr - 1 ranks; everybody else (Produce's substitution needs it on every rank, after the
    // sweep) gets it with the L panel exchange below, off the critical path.
    ops.clear();
    if (Pr > 1 && in_col) {
      if (is_diag) {
        for (int r2 = 0; r2 < Pr; ++r2)
          if (r2 != pr) ops.push_back(xop<T>(d->rank_of(r2, kc), true, Dv, (int64_t)nb2));
      } else {
        ops.push_back(xop<T>(diag_rank, false, Dv, (int64_t)nb2));
      }
      TRCHK(d->tr->group(sc, ops, &e_));
    }
    rec(h, E(P, ED), sc);

    // ---- 2. panel solve --------------------------------------------------------------------
    if (in_col) {
      wait(h, sp, E(P, ED));
      if (bi0 < mloc) {
        GemmGrid gtri;  // the tile inverse is lower triangular: tile column j only sums k < (j + 1) * 128
        gtri.ktri = h->ktri;
        launch_gemm_nt(sp, GEMM_RECT, (mloc - bi0) * tpb, tpb, nb, 1.0,
                        A + (size_t)bi0 * nb * ldA + (size_t)bj_d * nb, ldA, Dv, nb, 0.0,
                        d->lchunk<T>(bj_d) + (size_t)bi0 * nb2, nb, pf, &gtri);
      }
      rec(h, E(P, EPANEL), sp);
      wait(h, sc, E(P, EPANEL));
    }

    // ---- 3. exchange of the L panel -----------------------------------------------------------
    T *Lrow = in_col ? d->lchunk<T>(bj_d) : d->mat<T>(d->Lrow[lslot]);
    // square grid, diagonal rank: my tile columns are my tile rows -- no copy, the same buffer
    const bool alias = (q == 1 && pc % Pr == pr);
    T *Lcol = alias ? Lrow : d->mat<T>(d->Lcol[lslot]);
    if (P >= LRING)  // the panel buffers of step P - LRING are still being read by its updates
      for (int k : {EUPD, ELA, ERUPD, ERLA}) wait(h, sc, E(P - LRING, k));
    ops.clear();
    const int64_t cnt_row = (int64_t)(mloc - bi0) * (int64_t)nb2;
    if (in_col) {
      for (int c = 0; c < Pc; ++c)
        if (c != kc) ops.push_back(xop<T>(d->rank_of(pr, c), true, Lrow + (size_t)bi0 * nb2, cnt_row));
      if (Pr > 1) {
        int idx = 0;
        for (int pc2 = pr; pc2 < Pc; pc2 += Pr, ++idx) {  // receivers' grid columns served by me
          const int bj02 = first_gt(P, pc2, Pc);
          const int nblk = nloc - bj02;
          if (nblk <= 0) continue;
          T *pk = packb + (size_t)idx * nloc * nb2;
          pack_blocks_t<T>(sc, pk, Lrow, nblk, (int64_t)nb2, bj02 * q + (pc2 - pr) / Pr, q);
          for (int r2 = 0; r2 < Pr; ++r2)
            if (r2 != pr) ops.push_back(xop<T>(d->rank_of(r2, pc2), true, pk, (int64_t)nblk * (int64_t)nb2));
        }
      }
    } else {
      ops.push_back(xop<T>(d->rank_of(pr, kc), false, Lrow + (size_t)bi0 * nb2, cnt_row));
    }
    const int src_r = pc % Pr;  // grid row of the rank holding the tiles of my tile columns
    if (src_r != pr && nloc - bj0 > 0)
      ops.push_back(xop<T>(d->rank_of(src_r, kc), false, Lcol + (size_t)bj0 * nb2, (int64_t)(nloc - bj0) * (int64_t)nb2));
    // the diagonal tile's inverse for the ranks outside process column kc (last in every pair's list)
    if (is_diag) {
      for (int r = 0; r < d->nranks; ++r)
        if (r % Pc != kc) ops.push_back(xop<T>(r, true, Dv, (int64_t)nb2));
    } else if (!in_col) {
      ops.push_back(xop<T>(diag_rank, false, Dv, (int64_t)nb2));
    }
    TRCHK(d->tr->group(sc, ops, &e_));
    if (src_r == pr && !alias)
      pack_blocks_t<T>(sc, Lcol + (size_t)bj0 * nb2, Lrow, nloc - bj0, (int64_t)nb2,
                         bj0 * q + (pc - pr) / Pr, q);
    rec(h, E(P, EL), sc);

    // ---- 4. trailing update ------------------------------------------------------------------
    {
      GemmGrid gg;
      gg.rule = 1;
      gg.tpb_shift = d->tpb_shift;
      gg.pr = pr;
      gg.Pr = Pr;
      gg.pc = pc;
      gg.Pc = Pc;
      gg.rblk0 = bi0;
      int cst = bj0;
      if (next_mine && bi0 < mloc && bj0 < nloc) {
        // look-ahead: the next panel's block column (local block bj0) on the chain stream
        wait(h, sp, E(P, EL));
        if (P >= 1) wait(h, sp, E(P - 1, EUPD));  // bulk update of step P-1 touched it
        gg.cblk0 = bj0;
        launch_gemm_nt(sp, GEMM_RECT, (mloc - bi0) * tpb, tpb, nb, -1.0, Lrow + (size_t)bi0 * nb2, nb,
                        Lcol + (size_t)bj0 * nb2, nb, 1.0, A + (size_t)bi0 * nb * ldA + (size_t)bj0 * nb,
                        ldA, pf, &gg);
        cst = bj0 + 1;
      }
      // fp32 evaluation: my diagonal tiles below block row P take this panel's contribution in fp64 -- the one the next
      // step factors on the chain stream, the others with the bulk update
      int dt0 = 0, dtn = 0;   // first of my diagonal tiles with global index > P, and whether it is tile P + 1
      if (d->d64_on) {
        dt0 = bi0 <= d->dq_first ? 0 : (bi0 - d->dq_first + d->dq_step - 1) / d->dq_step;
        if (dt0 < d->dq_count && pr + Pr * (d->dq_first + dt0 * d->dq_step) == P + 1) {
          wait(h, sp, E(P, EL));
          if (P >= 1) wait(h, sp, E(P - 1, EUPD));
          d64_tiles_update(d, sp, Lrow, dt0, 1);
          dtn = 1;
        }
      }
      rec(h, E(P, ELA), sp);
      wait(h, s, E(P, EL));
      if (d->d64_on && dt0 + dtn < d->dq_count) d64_tiles_update(d, s, Lrow, dt0 + dtn, d->dq_count - dt0 - dtn);
      if (bi0 < mloc && cst < nloc) {
        gg.cblk0 = cst;
        launch_gemm_nt(s, GEMM_RECT, (mloc - bi0) * tpb, (nloc - cst) * tpb, nb, -1.0,
                        Lrow + (size_t)bi0 * nb2, nb, Lcol + (size_t)cst * nb2, nb, 1.0,
                        A + (size_t)bi0 * nb * ldA + (size_t)cst * nb, ldA, pf, &gg);
      }
      rec(h, E(P, EUPD), s);
    }
    return GOGP_OK;
  };

  // ---- inverse part of block column P (sections 5-7) ------------------------------------------------
  auto inv_step = [&](int P) -> int {
    const int kr = P % Pr, kc = P % Pc;
    const int bi_d = P / Pr, bj_d = P / Pc;
    const int bi0 = first_gt(P, pr, Pr);
    const int bj0 = first_gt(P, pc, Pc);
    const bool in_col = (pc == kc), in_row = (pr == kr), is_diag = in_col && in_row;
    const int slot = P & 1, lslot = P % LRING;
    T *Dv = Dinv + (size_t)P * nb2;
    const bool next_mine = (P + 1 < NB) && (pc == (P + 1) % Pc);
    const int src_r = pc % Pr;
    const bool alias = (q == 1 && src_r == pr);
    T *Lcol = alias ? (in_col ? d->lchunk<T>(bj_d) : d->mat<T>(d->Lrow[lslot])) : d->mat<T>(d->Lcol[lslot]);

    // ---- 5. column panel P of Y = L^-T -----------------------------------------------------------
    const int bim = in_row ? bi0 - 1 : bi0;  // local row blocks with global index < P
    if (in_col) {
      wait(h, st, E(P, ED));
      if (P >= 2) wait(h, st, E(P - 2, ERUPD));  // bulk R update of step P-2 touched column P
      if (bim > 0) {
        GemmGrid gtri;
        gtri.ktri = h->ktri;
        launch_gemm_nt(st, GEMM_RECT, bim * tpb, tpb, nb, 1.0, A + (size_t)bj_d * nb, ldA, Dv, nb, 0.0,
                        d->ychunk<T>(bj_d), nb, pf, &gtri);
      }
      if (is_diag) launch_transpose_sq(st, Dv, nb, d->ychunk<T>(bj_d) + (size_t)bi_d * nb2, nb, nb);
      if (bi0 > 0)  // z_P += sum_I Y[I, P]^T y_I over my tile rows
        launch_chunk_tdot(st, d->ychunk<T>(bj_d), (int64_t)bi0 * nb, nb, d->yloc, d->tpart, d->red + (size_t)P * nb);
      rec(h, E(P, EYCH), st);
      wait(h, sc, E(P, EYCH));
    }

    // ---- 6. exchange of the Y panel ------------------------------------------------------------
    T *Yrow = in_col ? d->ychunk<T>(bj_d) : d->mat<T>(d->Yrow[slot]);
    T *Ycol = alias ? Yrow : d->mat<T>(d->Ycol[slot]);
    if (P >= 2)  // the Y panel buffers of step P-2 are still being read by its updates
      for (int k : {ERUPD, ERLA}) wait(h, sc, E(P - 2, k));
    ops.clear();
    const int64_t cnt_yrow = (int64_t)bi0 * (int64_t)nb2;
    if (in_col) {
      for (int c = 0; c < Pc; ++c)
        if (c != kc) ops.push_back(xop<T>(d->rank_of(pr, c), true, Yrow, cnt_yrow));
      if (Pr > 1) {
        int idx = 0;
        for (int pc2 = pr; pc2 < Pc; pc2 += Pr, ++idx) {
          const int nblk = first_gt(P, pc2, Pc);  // tile columns <= P of grid column pc2
          if (nblk <= 0) continue;
          T *pk = packb + (size_t)idx * nloc * nb2;
          pack_blocks_t<T>(sc, pk, Yrow, nblk, (int64_t)nb2, (pc2 - pr) / Pr, q);
          for (int r2 = 0; r2 < Pr; ++r2)
            if (r2 != pr) ops.push_back(xop<T>(d->rank_of(r2, pc2), true, pk, (int64_t)nblk * (int64_t)nb2));
        }
      }
    } else {
      ops.push_back(xop<T>(d->rank_of(pr, kc), false, Yrow, cnt_yrow));
    }
    if (src_r != pr && bj0 > 0)
      ops.push_back(xop<T>(d->rank_of(src_r, kc), false, Ycol, (int64_t)bj0 * (int64_t)nb2));
    TRCHK(d->tr->group(sc, ops, &e_));
    if (src_r == pr && !alias) pack_blocks_t<T>(sc, Ycol, Yrow, bj0, (int64_t)nb2, (pc - pr) / Pr, q);
    rec(h, E(P, EY), sc);

    // ---- 7. R update (rows <= P, columns > P) and the rank-nb update of K^-1 -----------------------
    {
      int cst = bj0;
      if (next_mine && bi0 > 0 && bj0 < nloc) {
        wait(h, st, E(P, EY));
        wait(h, st, E(P, EL));
        if (P >= 1) wait(h, st, E(P - 1, ERUPD));
        launch_gemm_nt(st, GEMM_RECT, bi0 * tpb, tpb, nb, -1.0, Yrow, nb, Lcol + (size_t)bj0 * nb2, nb, 1.0,
                        A + (size_t)bj0 * nb, ldA, pf);
        cst = bj0 + 1;
      }
      rec(h, E(P, ERLA), st);
      wait(h, s2, E(P, EY));
      wait(h, s2, E(P, EL));
      if (bi0 > 0 && cst < nloc)
        launch_gemm_nt(s2, GEMM_RECT, bi0 * tpb, (nloc - cst) * tpb, nb, -1.0, Yrow, nb,
                        Lcol + (size_t)cst * nb2, nb, 1.0, A + (size_t)cst * nb, ldA, pf);
      if (want_kinv && bi0 > 0 && bj0 > 0) {
        GemmGrid gk;
        gk.rule = 2;
        gk.tpb_shift = d->tpb_shift;
        gk.pr = pr;
        gk.Pr = Pr;
        gk.pc = pc;
        gk.Pc = Pc;
        gk.rblk0 = 0;
        gk.cblk0 = 0;
        gk.beta0 = P;
        launch_gemm_nt(s2, GEMM_RECT, bi0 * tpb, bj0 * tpb, nb, 1.0, Yrow, nb, Ycol, nb, 1.0, A, ldA, pf,
                        &gk);
      }
      rec(h, E(P, ERUPD), s2);
    }
    return GOGP_OK;
  };

  for (int t = 0; t < NB + INV_LAG; ++t) {
    if (t < NB) {
      rc = chol_step(t);
      if (rc != GOGP_OK) return rc;
    }
    if (t >= INV_LAG) {
      rc = inv_step(t - INV_LAG);
      if (rc != GOGP_OK) return rc;
    }
  }

  // ---- z = Y^T y, log-determinant, failure flags: one all-reduce -------------------------------------
  rec(h, EV_FWD, st);
  wait(h, sc, EV_FWD);
  rec(h, EV_W, sp);
  wait(h, sc, EV_W);
  launch_info_to_double(sc, h->info, d->red + (size_t)npad + 1 + d->rank);
  TRCHK(d->tr->allreduce(sc, d->red, npad + 1 + d->nranks, &e_));
  HIPCHK(h, hipMemcpyAsync(h->z, d->red, (size_t)npad * sizeof(double), hipMemcpyDeviceToDevice, sc));
  launch_sumsq_info(sc, h->z, h->n, nullptr, h->scalars);
  // alpha = Y z: partial sums over the local chunks, second all-reduce
  launch_chunk_alpha(sc, d->mat<T>(d->Ych), mloc, nloc, nb, d->map(), h->z, d->ared);
  TRCHK(d->tr->allreduce(sc, d->ared, npad, &e_));
  HIPCHK(h, hipMemcpyAsync(h->alpha, d->ared, (size_t)npad * sizeof(double), hipMemcpyDeviceToDevice, sc));
  HIPCHK(h, hipMemcpyAsync(h->hscal + 1, d->red + npad, (size_t)(1 + d->nranks) * sizeof(double),
                           hipMemcpyDeviceToHost, sc));
  if (sizeof(T) == 4) {
    // fp32 tiles: `refine_steps` steps of iterative refinement of alpha against the EXACT Gram matrix
    // (api.hip: factorize_t; DESIGN.md "fp32 path"), sharded: every rank multiplies its share of the
    // columns of K (recomputed in fp64 on the fly) with alpha, one all-reduce gives K alpha; the
    // correction Y (Y^T r) uses the local chunks of Y like z and alpha above (two all-reduces).  The
    // quadratic term of the LML is y^T alpha of the refined alpha.
    for (int it = 0; it < h->refine_steps; ++it) {
      launch_kmatvec_share(sc, h->devP, h->D, h->dX, h->n, npad, h->alpha, d->rank, d->nranks, h->rpart,
                           REFINE_SLABS, h->rw, h->radial1);
      TRCHK(d->tr->allreduce(sc, h->rw, npad, &e_));
      launch_residual_from(sc, h->rw, h->dy, npad);  // rw := y - K alpha, then its local rows
      hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((lrows + 255) / 256)), dim3(256), 0, sc, h->rw,
                         d->rloc, lrows, d->map());
      // both reduction vectors start from zero: every rank fills only its own panels / rows
      HIPCHK(h, hipMemsetAsync(d->red, 0, (size_t)npad * sizeof(double), sc));
      HIPCHK(h, hipMemsetAsync(d->ared, 0, (size_t)npad * sizeof(double), sc));
      for (int bj = 0; bj < nloc; ++bj) {
        const int P = bj * Pc + pc;
        const int bi0 = first_gt(P, pr, Pr);
        if (bi0 > 0)
          launch_chunk_tdot(sc, d->ychunk<T>(bj), (int64_t)bi0 * nb, nb, d->rloc, d->tpart, d->red + (size_t)P * nb);
      }
      TRCHK(d->tr->allreduce(sc, d->red, npad, &e_));
      launch_chunk_alpha(sc, d->mat<T>(d->Ych), mloc, nloc, nb, d->map(), d->red, d->ared);
      TRCHK(d->tr->allreduce(sc, d->ared, npad, &e_));
      launch_axpy(sc, h->alpha, d->ared, npad);
    }
    launch_dot(sc, h->dy, h->alpha, h->n, h->scalars);
  }
  HIPCHK(h, hipMemcpyAsync(h->hscal, h->scalars, sizeof(double), hipMemcpyDeviceToHost, sc));
  HIPCHK(h, hipStreamSynchronize(sc));
  HIPCHK(h, hipGetLastError());
  const double zz = h->hscal[0], logdet = h->hscal[1];
  double first_fail = 0.0;
  for (int r = 0; r < d->nranks; ++r) {
    const double v = h->hscal[2 + r];
    if (v > 0.0 && (first_fail == 0.0 || v < first_fail)) first_fail = v;
  }
  if (first_fail != 0.0 || !std::isfinite(logdet) || !std::isfinite(zz)) {
    for (hipStream_t qs : {s, sp, st, s2}) (void)hipStreamSynchronize(qs);
    h->notpd = first_fail != 0.0 ? (int64_t)first_fail - 1 : -1;
    char buf[160];
    snprintf(buf, sizeof buf, "Factorize: matrix is not positive definite (pivot %lld)", (long long)h->notpd);
    h->err = buf;
    return GOGP_ENOTPD;
  }
  h->lml = -0.5 * (double)h->n * log(2 * M_PI) - 0.5 * logdet - 0.5 * zz;  // gp/gp.go:244-253
  h->yta = zz;  // y^T alpha (float tiles: of the refined alpha) -- the output-scale identity of the fp32 gradient
  h->factored = true;
  h->have_alpha = true;
  h->have_kinv = want_kinv;
  h->trtri_done = true;
  return GOGP_OK;
}

int gogp_dist_factorize(gogp_handle *h, bool want_kinv) {
  return h->prec == 32 ? dist_factorize_t<float>(h, want_kinv) : dist_factorize_t<double>(h, want_kinv);
}

// ---- gradient: fused reduction over the local tiles of K^-1, one all-reduce ------------------------
int gogp_dist_gradient_sums(gogp_handle *h, double *hacc) {
  Dist2D *d = h->dist;
  if (!h->have_kinv) return fail(h, GOGP_ESTATE, "Gradient: K^-1 was not formed (Absorb?)");
  hipStream_t s2 = h->s2, sc = d->sc;
  // the last rank-nb updates of K^-1 run on s2; alpha is final (host-synchronised)
  if (h->prec == 32)
    launch_grad_reduce_local(s2, h->devP, h->D, h->ard_dims, h->dX, h->alpha, d->mat<float>(d->A), d->ldA(), h->n,
                             (int64_t)d->mloc * d->nb, (int64_t)d->nloc * d->nb, d->map(), d->gpart, h->gout, h->radial1, h->ard_mfma_min);
  else
    launch_grad_reduce_local(s2, h->devP, h->D, h->ard_dims, h->dX, h->alpha, d->A, d->ldA(), h->n,
                             (int64_t)d->mloc * d->nb, (int64_t)d->nloc * d->nb, d->map(), d->gpart, h->gout, h->radial1, h->ard_mfma_min);
  // float tiles: tr(K^-1) = |Y|_F^2 and |alpha|^2 in fp64 ride along in two free slots (13, 14) of the all-reduce
  const bool tr64 = h->prec == 32 && h->trace_fp64 && d->rloc;
  if (tr64) {
    launch_chunk_sumsq(s2, d->mat<float>(d->Ych), d->mloc, d->nloc, d->nb, d->map(), h->n, d->rloc, h->gout + 13);
    launch_dot(s2, h->alpha, h->alpha, h->n, h->gout + 14);  // replicated: the all-reduce multiplies it by the ranks
  }
  rec(h, EV_ALPHA, s2);
  wait(h, sc, EV_ALPHA);
  TRCHK(d->tr->allreduce(sc, h->gout, NACC, &e_));
  HIPCHK(h, hipMemcpyAsync(hacc, h->gout, NACC * sizeof(double), hipMemcpyDeviceToHost, sc));
  HIPCHK(h, hipStreamSynchronize(sc));
  HIPCHK(h, hipGetLastError());
  if (tr64) {  // tr(alpha alpha^T - K^-1) from the fp64 sums over the chunks of Y instead of the float diagonal of K^-1
    hacc[ACC_TRACE] = hacc[14] / (double)d->nranks - hacc[13];
    hacc[13] = hacc[14] = 0.0;
  }
  return GOGP_OK;
}

// ---- gp.GP.L of a sharded handle: gather the tiles (collective; every rank gets the whole factor) ---
template <class T>
__global__ void scatter_tile_kernel(const T *__restrict__ src, int nb, double *__restrict__ dst, long n,
                                    long row0, long col0, int diag) {
  const long r = row0 + blockIdx.x;
  if (r >= n) return;
  for (int c = threadIdx.x; c < nb; c += blockDim.x) {
    const long gc = col0 + c;
    if (gc < n && (!diag || gc <= r)) dst[r * n + gc] = (double)src[(long)blockIdx.x * nb + c];
  }
}

// Collective agreement before a large collective: every rank contributes its own failure flag; if any
// rank failed (an allocation, a bad argument) ALL return the error instead of some of them blocking in the
// all-reduce that the failed rank never enters.  `flag`: 0 = fine.
static int agree_ok(gogp_handle *h, Dist2D *d, int flag, const char *what) {
  double *dflag = d->red;  // >= 1 double once data are set
  double hf = (double)(flag != 0);
  std::string terr;
  hipError_t e = hipMemcpyAsync(dflag, &hf, sizeof(double), hipMemcpyHostToDevice, d->sc);
  if (e == hipSuccess) e = hipStreamSynchronize(d->sc);
  int rc = (e == hipSuccess) ? d->tr->allreduce(d->sc, dflag, 1, &terr) : GOGP_EHIP;
  if (rc == GOGP_OK) {
    e = hipMemcpyAsync(&hf, dflag, sizeof(double), hipMemcpyDeviceToHost, d->sc);
    if (e == hipSuccess) e = hipStreamSynchronize(d->sc);
    if (e != hipSuccess) rc = GOGP_EHIP;
  }
  if (rc != GOGP_OK) {
    h->err = std::string(what) + ": " + (terr.empty() ? "HIP error while agreeing on the status" : terr);
    (void)hipGetLastError();
    return rc;
  }
  if (hf != 0.0) {
    if (!flag) h->err = std::string(what) + ": another rank failed before the collective";
    return flag ? flag : GOGP_ESTATE;
  }
  return GOGP_OK;
}

// The whole factor on every rank, gathered in ROW BANDS of bounded size (<= 256 MiB of device scratch
// per rank instead of one n x n buffer -- 34 GB at N = 65536, which would cancel the 1 / (Pr Pc) memory
// footprint of the shards): per band every rank writes the rows its tiles hold into the zeroed band, one
// all-reduce completes it, and it is copied to the host.
template <class T>
static int dist_get_factor_t(gogp_handle *h, double *Lout) {
  Dist2D *d = h->dist;
  const int nb = d->nb;
  const int64_t n = h->n;
  for (hipStream_t qs : {h->s, h->sp, h->st, h->s2, d->sc}) HIPCHK(h, hipStreamSynchronize(qs));
  int64_t band = ((int64_t)(32 << 20) / n) / nb * nb;  // rows per band: 32 Mi doubles, whole tile rows
  if (band < nb) band = nb;
  double *tmp = nullptr;
  const hipError_t ea = hipMalloc(&tmp, (size_t)band * n * sizeof(double));
  if (ea != hipSuccess) (void)hipGetLastError();
  int rc = agree_ok(h, d, ea == hipSuccess ? 0 : GOGP_ENOMEM, "sharded get_factor");
  if (rc != GOGP_OK) {
    (void)hipFree(tmp);
    if (ea != hipSuccess) h->err = "sharded get_factor: out of device memory";
    return rc;
  }
  hipError_t e = hipSuccess;
  std::string terr;
  for (int64_t r0 = 0; r0 < n && e == hipSuccess && rc == GOGP_OK; r0 += band) {
    const int64_t rows = std::min(band, n - r0);
    e = hipMemsetAsync(tmp, 0, (size_t)rows * n * sizeof(double), d->sc);
    for (int bi = 0; bi < d->mloc && e == hipSuccess; ++bi) {
      const int64_t gI = (int64_t)bi * d->Pr + d->pr;
      if (gI * nb < r0 || gI * nb >= r0 + rows) continue;  // bands hold whole tile rows
      for (int bj = 0; bj < d->nloc; ++bj) {
        const int64_t gP = (int64_t)bj * d->Pc + d->pc;
        if (gI < gP || gP * nb >= n) continue;
        // (the kernel's destination is the band: row index relative to r0)
        hipLaunchKernelGGL(scatter_tile_kernel<T>, dim3(nb), dim3(256), 0, d->sc,
                           d->lchunk<T>(bj) + (size_t)bi * nb * nb, nb, tmp - (size_t)r0 * n, (long)n, (long)gI * nb,
                           (long)gP * nb, gI == gP ? 1 : 0);
      }
    }
    if (e == hipSuccess) rc = d->tr->allreduce(d->sc, tmp, rows * n, &terr);
    if (e == hipSuccess && rc == GOGP_OK)
      e = hipMemcpyAsync(Lout + (size_t)r0 * n, tmp, (size_t)rows * n * sizeof(double), hipMemcpyDeviceToHost, d->sc);
    if (e == hipSuccess && rc == GOGP_OK) e = hipStreamSynchronize(d->sc);  // the band buffer is reused
  }
  (void)hipFree(tmp);
  if (rc != GOGP_OK) {
    h->err = "sharded get_factor: " + terr;
    return rc;
  }
  HIPCHK(h, e);
  return GOGP_OK;
}

int gogp_dist_get_factor(gogp_handle *h, double *Lout) {
  return h->prec == 32 ? dist_get_factor_t<float>(h, Lout) : dist_get_factor_t<double>(h, Lout);
}

// Selected rows of L (mode 0: out is nrows x n) or its diagonal (mode 1: rows = nullptr, out has n
// entries): every rank writes what its tiles hold into a zeroed device buffer, one all-reduce
// completes it on every rank (collective).
template <class T>
__global__ void gather_rows_of_l_kernel(const T *__restrict__ Lch, int mloc, int nloc, int nb, BlockMap map,
                                        const long *__restrict__ rows, long nrows, long n, int diag_only,
                                        double *__restrict__ out) {
  const long k = blockIdx.x;                       // requested row (or, diag_only, the row itself)
  const long r = diag_only ? k : rows[k];
  const int gI = (int)(r / nb);
  if (gI % map.Pr != map.pr) return;               // not one of my tile rows
  const int bi = gI / map.Pr;
  const long lr = r - (long)gI * nb;
  const long chunk_sz = (long)mloc * nb * nb;
  for (int bj = 0; bj < nloc; ++bj) {
    const int gP = bj * map.Pc + map.pc;
    if (gP > gI) break;
    const T *src = Lch + bj * chunk_sz + ((long)bi * nb + lr) * nb;
    if (diag_only) {
      if (gP == gI && threadIdx.x == 0) out[r] = (double)src[lr];
      continue;
    }
    for (int c = threadIdx.x; c < nb; c += blockDim.x) {
      const long gc = (long)gP * nb + c;
      if (gc < n && gc <= r) out[k * n + gc] = (double)src[c];
    }
  }
}

template <class T>
static int dist_get_factor_part_t(gogp_handle *h, const int64_t *rows, int64_t nrows, double *out) {
  Dist2D *d = h->dist;
  const int64_t n = h->n;
  const bool diag_only = rows == nullptr;
  const int64_t count = diag_only ? n : nrows * n, nblocks = diag_only ? n : nrows;
  for (hipStream_t qs : {h->s, h->sp, h->st, h->s2, d->sc}) HIPCHK(h, hipStreamSynchronize(qs));
  int bad = 0;
  for (int64_t k = 0; !diag_only && k < nrows; ++k)
    if (rows[k] < 0 || rows[k] >= n) bad = GOGP_EARG;  // the kernel indexes the chunks with them
  double *tmp = nullptr;
  long *drows = nullptr;
  if (!bad && hipMalloc(&tmp, (size_t)count * sizeof(double)) != hipSuccess) {
    (void)hipGetLastError();
    bad = GOGP_ENOMEM;
  }
  {
    const int rca = agree_ok(h, d, bad, "sharded get_factor_rows");
    if (rca != GOGP_OK) {
      (void)hipFree(tmp);
      if (bad == GOGP_EARG) h->err = "get_factor_rows: row out of range";
      if (bad == GOGP_ENOMEM) h->err = "sharded get_factor_rows: out of device memory";
      return rca;
    }
  }
  hipError_t e = hipMemsetAsync(tmp, 0, (size_t)count * sizeof(double), d->sc);
  if (e == hipSuccess && !diag_only) {
    e = hipMalloc(&drows, (size_t)nrows * sizeof(long));
    if (e == hipSuccess)
      e = hipMemcpyAsync(drows, rows, (size_t)nrows * sizeof(long), hipMemcpyHostToDevice, d->sc);
  }
  if (e == hipSuccess)
    hipLaunchKernelGGL(gather_rows_of_l_kernel<T>, dim3((unsigned)nblocks), dim3(256), 0, d->sc,
                       d->mat<T>(d->Lch), d->mloc, d->nloc, d->nb, d->map(), drows, (long)nrows, (long)n,
                       diag_only ? 1 : 0, tmp);
  int rc = GOGP_OK;
  std::string terr;
  if (e == hipSuccess) rc = d->tr->allreduce(d->sc, tmp, count, &terr);
  if (e == hipSuccess && rc == GOGP_OK)
    e = hipMemcpyAsync(out, tmp, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, d->sc);
  if (e == hipSuccess) e = hipStreamSynchronize(d->sc);
  (void)hipFree(tmp);
  (void)hipFree(drows);
  if (rc != GOGP_OK) {
    h->err = "sharded get_factor_rows: " + terr;
    return rc;
  }
  HIPCHK(h, e);
  return GOGP_OK;
}
int gogp_dist_get_factor_part(gogp_handle *h, const int64_t *rows, int64_t nrows, double *out) {
  static_assert(sizeof(long) == sizeof(int64_t), "row indices are copied as they are");
  return h->prec == 32 ? dist_get_factor_part_t<float>(h, rows, nrows, out)
                       : dist_get_factor_part_t<double>(h, rows, nrows, out);
}

// ---- gogp_set_factor on a sharded handle: "Produce on stored results" (gp/gp.go:255-257) ------------
// Collective; every rank passes the SAME dense n x n factor (what gogp_get_factor returns on every
// rank) and alpha.  Each rank keeps its own tiles of L (cut out of the host matrix one local tile row
// at a time through a staging buffer; rows / columns >= n get the identity padding of the padded
// matrix), inverts every diagonal tile itself (the NB tile inverses are replicated state) and stores
// alpha; no communication at all.  Y = L^-T is not rebuilt: Produce on the restored state runs the
// distributed forward substitution with L and the tile inverses (dist_produce_t).
template <class T>
__global__ void scatter_stage_kernel(const double *__restrict__ stage, long lds_, T *__restrict__ Lch, int nloc,
                                     int mloc, int bi, int nb) {
  // stage: nb rows x (nloc * nb) columns (this rank's tile columns of one local tile row)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)nb * nloc * nb;
  if (i >= total) return;
  const int r = (int)(i / ((long)nloc * nb));
  const long cc = i - (long)r * nloc * nb;
  const int bj = (int)(cc / nb), c = (int)(cc - (long)bj * nb);
  Lch[((size_t)bj * mloc + bi) * nb * nb + (size_t)r * nb + c] = (T)stage[(long)r * lds_ + cc];
}

template <class T>
static int dist_set_factor_t(gogp_handle *h, const double *Lin, const double *alpha) {
  Dist2D *d = h->dist;
  const int nb = d->nb, NB = d->NB, mloc = d->mloc, nloc = d->nloc, Pr = d->Pr, Pc = d->Pc, pr = d->pr, pc = d->pc;
  const int64_t n = h->n, npad = h->npad;
  const size_t nb2 = (size_t)nb * nb;
  hipStream_t s = h->s;
  for (hipStream_t qs : {h->s, h->sp, h->st, h->s2, h->sl, d->sc}) HIPCHK(h, hipStreamSynchronize(qs));
  int rc = gogp_upload_params(h);
  if (rc != GOGP_OK) return rc;
  // ---- my tiles of L ------------------------------------------------------------------------------
  const size_t stage_elems = (size_t)nb * nloc * nb;
  double *hst = nullptr, *dst = nullptr;
  HIPCHK(h, hipHostMalloc((void **)&hst, stage_elems * sizeof(double), hipHostMallocDefault));
  hipError_t e = hipMalloc(&dst, stage_elems * sizeof(double));
  if (e != hipSuccess) {
    (void)hipHostFree(hst);
    HIPCHK(h, e);
  }
  for (int bi = 0; bi < mloc && e == hipSuccess; ++bi) {
    const int64_t gI = (int64_t)bi * Pr + pr;
    for (int r = 0; r < nb; ++r) {
      const int64_t gi = gI * nb + r;
      double *row = hst + (size_t)r * nloc * nb;
      for (int bj = 0; bj < nloc; ++bj) {
        const int64_t gc0 = ((int64_t)bj * Pc + pc) * nb;
        double *dstp = row + (size_t)bj * nb;
        for (int c = 0; c < nb; ++c) {
          const int64_t gc = gc0 + c;
          double v = 0.0;
          if (gc <= gi) v = (gi < n) ? Lin[(size_t)gi * n + gc] : (gc == gi ? 1.0 : 0.0);
          dstp[c] = v;
        }
      }
    }
    e = hipMemcpyAsync(dst, hst, stage_elems * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(scatter_stage_kernel<T>, dim3((unsigned)((stage_elems + 255) / 256)), dim3(256), 0, s, dst,
                         (long)nloc * nb, d->mat<T>(d->Lch), nloc, mloc, bi, nb);
      e = hipStreamSynchronize(s);  // the staging buffer is refilled by the next tile row
    }
  }
  // ---- every diagonal tile's inverse, on every rank (fp64, rounded once on a float shard) -------------------
  double *t64 = nullptr;  // [L tile | inverse] in fp64
  if (e == hipSuccess) e = hipMalloc(&t64, 2 * nb2 * sizeof(double));
  double logdet = 0.0;
  for (int P = 0; P < NB && e == hipSuccess; ++P) {
    for (int r = 0; r < nb; ++r) {
      const int64_t gi = (int64_t)P * nb + r;
      for (int c = 0; c < nb; ++c) {
        const int64_t gc = (int64_t)P * nb + c;
        hst[(size_t)r * nb + c] = (gc > gi) ? 0.0 : ((gi < n) ? Lin[(size_t)gi * n + gc] : (gc == gi ? 1.0 : 0.0));
      }
      if (gi < n) logdet += 2.0 * log(Lin[(size_t)gi * n + gi]);
    }
    e = hipMemcpyAsync(t64, hst, nb2 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) break;
    double *Lt = t64, *Dv = t64 + nb2;
    launch_diag256_inv_only_ld512(s, Lt, nb, Dv);
    launch_diag256_inv_only_ld512(s, Lt + (size_t)PANEL * nb + PANEL, nb, Dv + (size_t)PANEL * nb + PANEL);
    tile_inverse_offdiag(d, s, Lt + (size_t)PANEL * nb, Dv, nullptr);
    launch_convert_block(s, Dv, nb, d->mat<T>(d->Dinv) + (size_t)P * nb2, nb, nb, nb);
    e = hipStreamSynchronize(s);
  }
  (void)hipFree(t64);
  (void)hipFree(dst);
  (void)hipHostFree(hst);
  HIPCHK(h, e);
  // ---- alpha and the LML of the restored state: -n/2 log 2 pi - sum log L_ii - 1/2 y^T alpha ------------------
  HIPCHK(h, hipMemsetAsync(h->alpha, 0, (size_t)npad * sizeof(double), s));
  HIPCHK(h, hipMemcpyAsync(h->alpha, alpha, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  launch_dot(s, h->dy, h->alpha, n, h->scalars);
  HIPCHK(h, hipMemcpyAsync(h->hscal, h->scalars, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  HIPCHK(h, hipGetLastError());
  h->lml = -0.5 * (double)n * log(2 * M_PI) - 0.5 * logdet - 0.5 * h->hscal[0];
  h->factored = true;
  h->have_alpha = true;
  h->alpha_pending = false;
  h->have_kinv = false;
  h->trtri_done = false;  // no Y: Produce substitutes with L
  return GOGP_OK;
}
int gogp_dist_set_factor(gogp_handle *h, const double *Lin, const double *alpha) {
  return h->prec == 32 ? dist_set_factor_t<float>(h, Lin, alpha) : dist_set_factor_t<double>(h, Lin, alpha);
}

// ---- Produce on a sharded handle -----------------------------------------------------------------
// gp.GP.Produce (gp/gp.go:258-360) with L never leaving its ranks:  sigma_j^2 = k(z_j, z_j) -
// |L^-1 k*_j|^2 and L^-1 = Y^T, so with V = Y^T Kstar:  V[P-block, j] = sum_{I <= P} Y[I, P]^T
// Kstar[I, j].  Every rank forms the partial sums of its own tiles (its chunks of Y against the
// cross-covariance rows of its tile rows: one NT GEMM per chunk against the transposed chunk),
// the Pr ranks of a process column add their partials (one exchange), column norms and
// mu = Kstar^T alpha partial sums meet in one all-reduce of 2 m doubles.
__global__ void gather_x_kernel(const double *__restrict__ X, const double *__restrict__ a, int D, long n,
                                long rows, BlockMap map, double *__restrict__ Xloc,
                                double *__restrict__ aloc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const long g = map.grow(i);
  for (int k = 0; k < D; ++k) Xloc[i * D + k] = (g < n) ? X[g * D + k] : 0.0;
  aloc[i] = (g < n) ? a[g] : 0.0;
}

// dst (cols x rows, ldd) = src (rows x cols, lds)^T; rows, cols multiples of 32
template <class T>
__global__ __launch_bounds__(256) void transpose_rect_kernel(const T *__restrict__ src, long lds_,
                                                             T *__restrict__ dst, long ldd, int ntc) {
  __shared__ T tile[32][33];
  const int bx = blockIdx.x % ntc, by = blockIdx.x / ntc;  // bx: column tile, by: row tile of src
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 32; r += 8) tile[ty + r][tx] = src[(long)(by * 32 + ty + r) * lds_ + bx * 32 + tx];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 32; r += 8) dst[(long)(bx * 32 + ty + r) * ldd + by * 32 + tx] = tile[tx][ty + r];
}

template <class T>
__global__ void add_inplace_kernel(T *__restrict__ a, const T *__restrict__ b, long count) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
    a[i] += b[i];
}

__global__ void scale_kernel(double *__restrict__ a, double f, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) a[i] *= f;
}

// dst (rows x nb, contiguous) = src (rows x nb block of a matrix with leading dimension ld)
template <class T>
__global__ void copy_block_kernel(T *__restrict__ dst, const T *__restrict__ src, long ld, int nb, long rows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nb) return;
  const long r = i / nb;
  const int c = (int)(i - r * nb);
  dst[i] = src[r * ld + c];
}
// W (rows x nb) = Ks block - U block - sum of nrecv received partial blocks (U and the partials fp64)
template <class T>
__global__ void residual_block_kernel(T *__restrict__ W, const T *__restrict__ Ks, const double *__restrict__ U,
                                      long ld, const double *__restrict__ recv, int nrecv, int nb, long rows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nb) return;
  const long r = i / nb;
  const int c = (int)(i - r * nb);
  double u = U[r * ld + c];
  for (int k = 0; k < nrecv; ++k) u += recv[(long)k * rows * nb + i];
  W[i] = (T)((double)Ks[r * ld + c] - u);
}
__global__ void add_vec_kernel(double *__restrict__ a, const double *__restrict__ b, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) a[i] += b[i];
}

// Float shards: V = L^-1 Kstar by a distributed blocked forward substitution with the float L tiles
// and the (fp64-computed) tile inverses, instead of V = Y^T Kstar with the explicit fp32 inverse --
// sigma^2 = prior - |V|^2 cancels, and the substitution's error does not grow with cond(L).
// Right-looking over block columns Q.  Every rank keeps U^T (mpad x local rows): the part of
// sum_{Q' < I} L[I, Q'] V_Q' that its own tile columns contribute -- ACCUMULATED IN FP64 (the float
// tiles and V_Q are widened per step and multiplied on the fp64 tile kernel): the single-GPU path
// updates its residual in place, so every rounding is relative to the shrinking residual; a sum kept
// apart from Kstar must not be rounded relative to its own (Kstar-sized) magnitude, or
// W_Q = Kstar_Q - sum loses what the cancellation in sigma^2 needs (measured: sigma 1e-2 with a float
// sum, the level of the explicit inverse).  Step Q:
//   a. the ranks of process row Q mod Pr send their block Q of U to the owner of tile (Q, Q), which
//      forms W_Q = Kstar_Q - sum U and V_Q = inv(L_QQ) W_Q (one GEMM with the replicated inverse)
//      and adds |V_Q[:, j]|^2 to q_j;
//   b. the owner sends V_Q to the other ranks of process column Q mod Pc;
//   c. those ranks add L[I, Q] V_Q to U for their tile rows I > Q (one GEMM).
// qsum (mpad doubles, zeroed by the caller) receives this rank's share of q; the caller all-reduces it.
template <class T>
static int produce_v_by_substitution(gogp_handle *h, Dist2D *d, const T *KsT, int64_t lrows, int64_t mpad, int64_t m,
                                     double *qsum, double *qtmp, hipStream_t s, hipStream_t sc) {
  const int nb = d->nb, Pr = d->Pr, Pc = d->Pc, pr = d->pr, pc = d->pc, mloc = d->mloc, NB = d->NB;
  const size_t nb2 = (size_t)nb * nb, blk = (size_t)mpad * nb;
  const int mt = (int)(mpad / TILE);
  T *Wq = nullptr, *Vq = nullptr;
  double *UT = nullptr, *recv = nullptr, *sendb = nullptr, *Vd = nullptr, *Ld = nullptr;
  auto cleanup = [&]() {
    for (T *p : {Wq, Vq}) (void)hipFree(p);
    for (double *p : {UT, recv, sendb, Vd, Ld}) (void)hipFree(p);
  };
  hipError_t e = hipMalloc(&UT, (size_t)mpad * lrows * sizeof(double) + 16);
  if (e == hipSuccess) e = hipMalloc(&Wq, blk * sizeof(T) + 16);
  if (e == hipSuccess) e = hipMalloc(&Vq, blk * sizeof(T) + 16);
  if (e == hipSuccess) e = hipMalloc(&recv, blk * (size_t)std::max(1, Pc - 1) * sizeof(double) + 16);
  if (e == hipSuccess) e = hipMalloc(&sendb, blk * sizeof(double) + 16);
  if (e == hipSuccess) e = hipMalloc(&Vd, blk * sizeof(double) + 16);
  if (e == hipSuccess) e = hipMalloc(&Ld, (size_t)mloc * nb2 * sizeof(double) + 16);
  if (e == hipSuccess) e = hipMemsetAsync(UT, 0, (size_t)mpad * lrows * sizeof(double), s);
  if (e != hipSuccess) {
    cleanup();
    (void)hipGetLastError();
    return fail(h, GOGP_ENOMEM, "Produce: out of device memory");
  }
  const T *Dinv = d->mat<T>(d->Dinv);
  const unsigned cb = (unsigned)((blk + 255) / 256);
  std::vector<XferOp> ops;
  std::string terr;
  int rc = GOGP_OK;
  for (int Q = 0; Q < NB && rc == GOGP_OK; ++Q) {
    const int kr = Q % Pr, kc = Q % Pc;
    const bool in_row = pr == kr, in_col = pc == kc, owner = in_row && in_col;
    const int bi = Q / Pr, bj = Q / Pc;
    // ---- a. partial sums of block Q to the owner
    ops.clear();
    if (in_row && !owner) {
      hipLaunchKernelGGL(copy_block_kernel<double>, dim3(cb), dim3(256), 0, s, sendb, UT + (size_t)bi * nb,
                         (long)lrows, nb, (long)mpad);
      rec(h, E(Q, 0), s);
      wait(h, sc, E(Q, 0));
      ops.push_back(xop<double>(d->rank_of(pr, kc), true, sendb, (int64_t)blk));
    } else if (owner) {
      int k = 0;
      for (int c = 0; c < Pc; ++c)
        if (c != kc) ops.push_back(xop<double>(d->rank_of(pr, c), false, recv + (size_t)(k++) * blk, (int64_t)blk));
    }
    rc = d->tr->group(sc, ops, &terr);
    if (rc != GOGP_OK) break;
    if (in_row && !owner) {  // sendb is packed again at a later step: not before this send is through
      rec(h, E(Q, 5), sc);
      wait(h, s, E(Q, 5));
    }
    if (owner) {
      rec(h, E(Q, 1), sc);
      wait(h, s, E(Q, 1));
      hipLaunchKernelGGL(residual_block_kernel<T>, dim3(cb), dim3(256), 0, s, Wq, KsT + (size_t)bi * nb,
                         UT + (size_t)bi * nb, (long)lrows, recv, Pc - 1, nb, (long)mpad);
      // V_Q^T (mpad x nb) = W_Q^T inv(L_QQ)^T
      launch_gemm_nt(s, GEMM_RECT, mt, d->tpb, nb, 1.0, Wq, nb, Dinv + (size_t)Q * nb2, nb, 0.0, Vq, nb, nullptr);
      launch_rownorm_dot(s, Vq, nb, nullptr, nb, m, nullptr, qtmp);
      hipLaunchKernelGGL(add_vec_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, qsum, qtmp, (long)m);
      rec(h, E(Q, 2), s);
      wait(h, sc, E(Q, 2));
    }
    // ---- b. V_Q down the process column
    ops.clear();
    if (owner) {
      for (int r2 = 0; r2 < Pr; ++r2)
        if (r2 != pr) ops.push_back(xop<T>(d->rank_of(r2, pc), true, Vq, (int64_t)blk));
    } else if (in_col) {
      ops.push_back(xop<T>(d->rank_of(kr, kc), false, Vq, (int64_t)blk));
    }
    rc = d->tr->group(sc, ops, &terr);
    if (rc != GOGP_OK) break;
    if (owner && Pr > 1) {  // Vq is written again (on s) when this rank owns a later step
      rec(h, E(Q, 6), sc);
      wait(h, s, E(Q, 6));
    }
    // ---- c. U[I] += L[I, Q] V_Q for my tile rows I > Q
    if (in_col) {
      const int bi0 = first_gt(Q, pr, Pr);
      if (!owner) {
        rec(h, E(Q, 3), sc);
        wait(h, s, E(Q, 3));
      }
      if (bi0 < mloc) {
        launch_convert_block(s, Vq, nb, Vd, nb, (int)mpad, nb);
        launch_convert_block(s, d->lchunk<T>(bj) + (size_t)bi0 * nb2, nb, Ld, nb, (mloc - bi0) * nb, nb);
        launch_dgemm_nt(s, GEMM_RECT, mt, (mloc - bi0) * d->tpb, nb, 1.0, Vd, nb, Ld, nb, 1.0, UT + (size_t)bi0 * nb,
                        lrows, nullptr);
      }
      // Vq / sendb are reused by the next step: its transfers wait for this update
      rec(h, E(Q, 4), s);
      wait(h, sc, E(Q, 4));
    }
  }
  if (rc != GOGP_OK) {
    (void)hipStreamSynchronize(sc);
    (void)hipStreamSynchronize(s);
    cleanup();
    h->err = "sharded Produce: " + terr;
    return rc;
  }
  e = hipStreamSynchronize(sc);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  cleanup();
  HIPCHK(h, e);
  return GOGP_OK;
}

template <class T>
static int dist_produce_t(gogp_handle *h, const double *Z, int64_t m, double *mu, double *sigma) {
  Dist2D *d = h->dist;
  if (!h->factored) return fail(h, GOGP_ESTATE, "Produce: no factorisation");
  const int nb = d->nb, Pr = d->Pr, Pc = d->Pc, pr = d->pr, pc = d->pc, mloc = d->mloc, nloc = d->nloc;
  const int64_t mpad = ((m + TILE - 1) / TILE) * TILE;
  const int64_t lrows = (int64_t)mloc * nb, lcols = (int64_t)nloc * nb;
  hipStream_t s = h->s, sc = d->sc;
  for (hipStream_t qs : {h->s, h->sp, h->st, h->s2, sc}) HIPCHK(h, hipStreamSynchronize(qs));
  double *dZ = nullptr, *Xloc = nullptr, *aloc = nullptr, *vec = nullptr;
  T *Ks = nullptr, *Yt = nullptr, *Vt = nullptr, *Vr = nullptr;  // matrices in the handle's element type
  auto cleanup = [&]() {
    for (double *p : {dZ, Xloc, aloc, vec}) (void)hipFree(p);
    for (T *p : {Ks, Yt, Vt, Vr}) (void)hipFree(p);
  };
#define PMALLOC(ptr, count)                                                              \
  do {                                                                                   \
    if (hipMalloc(&(ptr), (size_t)(count) * sizeof(*(ptr)) + 16) != hipSuccess) {        \
      cleanup();                                                                         \
      (void)hipGetLastError();                                                           \
      return fail(h, GOGP_ENOMEM, "Produce: out of device memory");                      \
    }                                                                                    \
  } while (0)
  PMALLOC(dZ, m * h->D);
  PMALLOC(Xloc, lrows * h->D);
  PMALLOC(aloc, lrows);
  PMALLOC(Ks, mpad * lrows);
  // float shards, and any shard whose factor was restored (gogp_set_factor: there is no Y = L^-T):
  // see produce_v_by_substitution
  const bool by_substitution = sizeof(T) == 4 || !h->trtri_done;
  if (!by_substitution) {
    PMALLOC(Yt, (int64_t)nb * lrows);
    PMALLOC(Vt, mpad * lcols);
    if (Pr > 1) PMALLOC(Vr, (int64_t)(Pr - 1) * mpad * lcols);
  }
  PMALLOC(vec, 5 * mpad);
  double *prior = vec, *red2 = vec + mpad /* [mu | q] */, *dsig = vec + 3 * mpad, *qtmp = vec + 4 * mpad;
  hipError_t e = hipMemcpyAsync(dZ, Z, (size_t)m * h->D * sizeof(double), hipMemcpyHostToDevice, s);
  if (e == hipSuccess && !by_substitution) e = hipMemsetAsync(Vt, 0, (size_t)mpad * lcols * sizeof(T), s);
  if (e == hipSuccess) e = hipMemsetAsync(red2, 0, (size_t)2 * mpad * sizeof(double), s);
  if (e != hipSuccess) {
    cleanup();
    HIPCHK(h, e);
  }
  launch_prior(s, h->devP, dZ, m, prior);  // gp/gp.go:269-278
  hipLaunchKernelGGL(gather_x_kernel, dim3((unsigned)((lrows + 255) / 256)), dim3(256), 0, s, h->dX,
                     h->alpha, h->D, (long)h->n, (long)lrows, d->map(), Xloc, aloc);
  // local rows are in increasing global order: the valid ones (global row < n) form a prefix
  int64_t nvalid = 0;
  {
    BlockMap mp = d->map();
    for (int64_t i = 0; i < lrows; i += nb) {
      const int64_t g0 = mp.grow(i);
      if (g0 >= h->n) break;
      nvalid += (h->n - g0 < nb) ? h->n - g0 : nb;
    }
  }
  // cross-covariance of the test points with my tile rows: Ks[j][lrow] (gp/gp.go:322-332)
  launch_cross(s, h->devP, h->D, Xloc, nvalid, lrows, dZ, m, mpad, Ks, lrows);
  launch_rownorm_dot(s, Ks, lrows, aloc, lrows, m, red2, nullptr);  // partial mu = Kstar^T alpha (:335)
  if (Pc > 1)  // the Pc ranks of a process row hold the same tile rows
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((mpad + 255) / 256)), dim3(256), 0, s, red2, 1.0 / Pc,
                       (long)mpad);
  const int mt = (int)(mpad / TILE);
  if (by_substitution) {
    const int rcs = produce_v_by_substitution<T>(h, d, Ks, lrows, mpad, m, red2 + mpad, qtmp, s, sc);
    if (rcs != GOGP_OK) {
      cleanup();
      return rcs;
    }
  }
  for (int bj = 0; !by_substitution && bj < nloc; ++bj) {
    const int P = bj * Pc + pc;
    const int bi0 = first_gt(P, pr, Pr);
    if (bi0 <= 0) continue;
    const int64_t K = (int64_t)bi0 * nb;
    hipLaunchKernelGGL(transpose_rect_kernel<T>, dim3((unsigned)((K / 32) * (nb / 32))), dim3(256), 0, s,
                       d->ychunk<T>(bj), (long)nb, Yt, (long)K, nb / 32);
    launch_gemm_nt(s, GEMM_RECT, mt, d->tpb, K, 1.0, Ks, lrows, Yt, K, 0.0, Vt + (size_t)bj * nb, lcols,
                   nullptr);
  }
  // add the partial sums of the other ranks of my process column
  std::vector<XferOp> ops;
  if (Pr > 1 && !by_substitution) {
    rec(h, EV_W, s);
    wait(h, sc, EV_W);
    int k = 0;
    for (int r2 = 0; r2 < Pr; ++r2) {
      if (r2 == pr) continue;
      ops.push_back(xop<T>(d->rank_of(r2, pc), true, Vt, mpad * lcols));
      ops.push_back(xop<T>(d->rank_of(r2, pc), false, Vr + (size_t)k * mpad * lcols, mpad * lcols));
      ++k;
    }
    std::string terr;
    int rc = d->tr->group(sc, ops, &terr);
    if (rc != GOGP_OK) {
      (void)hipStreamSynchronize(sc);
      cleanup();
      h->err = "sharded Produce: " + terr;
      return rc;
    }
    rec(h, EV_FWD, sc);
    wait(h, s, EV_FWD);
    for (int i = 0; i < Pr - 1; ++i)
      hipLaunchKernelGGL(add_inplace_kernel<T>, dim3(1024), dim3(256), 0, s, Vt, Vr + (size_t)i * mpad * lcols,
                         (long)(mpad * lcols));
  }
  // |V_j|^2 over my tile columns; every rank of a process column holds the same sums
  if (!by_substitution) launch_rownorm_dot(s, Vt, lcols, nullptr, lcols, m, nullptr, red2 + mpad);
  if (Pr > 1 && !by_substitution)
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((mpad + 255) / 256)), dim3(256), 0, s, red2 + mpad,
                       1.0 / Pr, (long)mpad);
  rec(h, EV_W, s);
  wait(h, sc, EV_W);
  {
    std::string terr;
    int rc = d->tr->allreduce(sc, red2, 2 * mpad, &terr);
    if (rc != GOGP_OK) {
      (void)hipStreamSynchronize(sc);
      cleanup();
      h->err = "sharded Produce: " + terr;
      return rc;
    }
  }
  launch_sigma(sc, prior, red2 + mpad, m, dsig);  // gp/gp.go:354-357, unclamped
  e = hipMemcpyAsync(mu, red2, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, sc);
  if (e == hipSuccess) e = hipMemcpyAsync(sigma, dsig, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, sc);
  if (e == hipSuccess) e = hipStreamSynchronize(sc);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  cleanup();
  HIPCHK(h, e);
  HIPCHK(h, hipGetLastError());
  return GOGP_OK;
}

int gogp_dist_produce(gogp_handle *h, const double *Z, int64_t m, double *mu, double *sigma) {
  return h->prec == 32 ? dist_produce_t<float>(h, Z, m, mu, sigma) : dist_produce_t<double>(h, Z, m, mu, sigma);
}
