// handle.h -- the handle behind the C ABI, shared by api.hip (single-GPU orchestration)
// and dist2d.hip (2-D block-cyclic sharded evaluation).  Private to libgogp_hip.so.
#pragma once
#include <algorithm>
#include <array>
#include <string>
#include <vector>

#include "common.h"

namespace gogp {
struct Dist2D;  // dist2d.hip
}
using gogp::DevParams;
using gogp::GemmProfile;

constexpr int REFINE_SLABS = 8;  // column slabs of the residual kernel (gram.hip: launch_residual)

struct gogp_handle {
  gogp_desc desc;
  int device = 0;
  int ns = 0, nn = 0, P = 0, D = 0;
  int ard_dims = 0;
  bool radial1 = false;  // the similarity kernel is one radial (non-periodic) term: restructured O(N^2) kernels
  int64_t n = 0, npad = 0;
  int nblk = 0;  // 128-blocks
  // device buffers
  double *dX = nullptr, *dy = nullptr;
  double *bufA = nullptr, *bufL = nullptr, *bufY = nullptr, *Dinv = nullptr;
  double *z = nullptr, *w = nullptr, *alpha = nullptr;
  double *scalars = nullptr;  // 8 doubles
  double *dscr = nullptr;     // fp32 path: fp64 scratch of the diagonal-block kernel (3 x 256 x 256)
  double *D64 = nullptr;      // fp32 path: the diagonal blocks accumulated in fp64 (diagsyrk.hip), npad / 256 blocks of 256 x 256
  int64_t cap_d64 = 0;        // ... allocated for this npad
  int diag_fp64 = 1;          // option "diag_fp64": 1 (default) the fp32 path's pivots come from that strip, 0 from the float matrix
  bool d64_active = false;    // set by the factorisation in progress
  int prec = 64;              // 64: fp64 throughout; 32: N x N matrices and O(N^3) products in fp32
  size_t esz() const { return prec == 32 ? sizeof(float) : sizeof(double); }
  int refine_steps = 1;       // fp32 path: iterative-refinement steps of alpha against the fp64 K
  double *rw = nullptr, *rz = nullptr, *rd = nullptr, *rpart = nullptr;  // its scratch
  long long *info = nullptr;
  double *gpart = nullptr, *gout = nullptr;
  DevParams *devP = nullptr;
  DevParams *hostP = nullptr;  // pinned
  double *hscal = nullptr;     // pinned staging, NACC + 16 doubles
  int64_t cap_npad = 0;        // allocation size of the N-dependent buffers
  int64_t cap_y = 0;           // npad bufY was allocated for (it is allocated lazily)
  // produce workspace
  double *dZ = nullptr, *KsT = nullptr, *Vt = nullptr, *pvec = nullptr;
  int64_t cap_m = 0, cap_mp_npad = 0;
  hipStream_t s = nullptr;   // main stream: Gram, big trailing updates, reductions
  hipStream_t sp = nullptr;  // panel stream (high priority): diagonal blocks, TRSM-as-GEMM,
                             // skinny updates, substitution steps -- overlaps the big updates
  // candidate batching (gogp_observe_gradient_candidates): k arena slots, each with its own copy of
  // every per-candidate buffer at the same offset (common.h: Batch); the handle's own buffers are
  // not touched by a batched evaluation
  char *cand_arena = nullptr;
  size_t cand_stride = 0;       // bytes per slot
  int cand_cap_k = 0;
  int64_t cand_cap_npad = 0;
  DevParams *cand_hostP = nullptr;  // pinned, cand_host_k entries
  double *cand_hscal = nullptr;     // pinned, cand_host_k x (NACC + 16) doubles
  int cand_host_k = 0;
  int batch_k = 1;              // > 1 only while a batched evaluation is being enqueued
  bool batch_mode = false;      // a batched evaluation is being enqueued (also with k = 1)
  // option "graph": the launch sequence of a batched evaluation captured once into a hipGraph and
  // replayed (the parameters change in pinned host memory only)
  int use_graph = 1;            // 1: linear graph from stream capture (N <= 1024; the default), 2: explicitly built DAG (graphrec.h), 3: the same recorder as one chain in enqueue order (diagnostics), 0: none
  bool graph_failed = false;    // the runtime refused the explicit graph once: stream path from then on
  int graph_nodes = 0;          // nodes of the graph in use (diagnostics)
  std::string graph_note;
  hipGraphExec_t cand_graph = nullptr;
  hipStream_t sg = nullptr;     // capture / replay stream of that graph (created on first use)
  struct {
    int k = 0, superpanel = 0, kinv_fused = 0;
    int64_t n = 0, cap_npad = 0;
    size_t stride = 0;
    const void *arena = nullptr, *dX = nullptr, *dy = nullptr, *hostP = nullptr, *hscal = nullptr;
  } cand_graph_key, cand_seen_key;  // what the graph was captured for / what the last call asked for
  // sharded evaluation (gogp_dist_init_*): 2-D block-cyclic state, nullptr on a single GPU
  gogp::Dist2D *dist = nullptr;
  hipStream_t sl = nullptr;  // forward substitution steps (low priority, off the chain)
  hipStream_t s2 = nullptr;  // big updates of the triangular inverse (fused sweep)
  hipStream_t st = nullptr;  // its chain: column panels of Y = L^-T
  hipStream_t sk = nullptr;  // rank-k updates K^-1 (+)= Y_P Y_P^T behind the inverse (low priority)
  void *stream_set = nullptr;  // the pooled StreamSet the five streams belong to
  std::vector<hipEvent_t> evs;  // cross-stream ordering events (timing disabled)
  int lookahead = 1;
  int superpanel = 2;          // 256-wide panels per trailing update (K = 256*superpanel)
  int eager = 1;               // Observe also runs the triangular inverse (gradient
                               // preparation), interleaved with the Cholesky sweep
  int superpanel_head = -1;    // > 0: super-panel width while more than head_remaining panels are to come; -1: 3 (fp64) / 4 (fp32)
  int head_remaining = 16;     // (measured, N = 16384: 3 / 16 72.7 ms, 3 / 24 72.8, 4 / 32 74.2, off 73.0-73.4)
  int chain_prio = -1;         // tile-kernel launches on the two chains raise their waves' issue priority (-1: by size)
  int64_t chain_tail = 0;      // chain_split = -1, large N beside the inverse: panel128 for the super-panels with at most this many rows left (0: none)
  int tiny = 1;                // N <= 128 observations: one launch for the whole factorisation (api.hip: tiny_factorize)
  int chain_slabs = 0;         // chain_split = 2: slabs of 64 panel rows per workgroup (panel128.hip); 0: by the launch's size
  int chain_split = -1;        // 1: the diagonal block in two 128-halves, their products on the tile kernel; 2: panel128.hip (api.hip; -1: by size)
  int ktri = 1;                // panel solves skip the zero half of the block inverse (common.h: GemmGrid)
  // T^-1 (lower, row-major) of the diagonal block of every super-panel of the factor: rows C0.. of an npad x tinv_ld
  // matrix, assembled by the first Produce on a factor (api.hip: assemble_tinv, produce_solve_t).  Produce then solves
  // a super-panel of columns with ONE product instead of a solve + update per 256 columns.
  double *TX = nullptr, *Tmt = nullptr;  // Tmt: 256 x 256 scratch blocks of the assembly
  int64_t tinv_ld = 0;
  size_t cap_tinv = 0;                   // elements allocated for TX
  // option "gradient_precision" = 32 on an fp64 handle: the factorisation, LML, alpha and Produce stay fp64; only what
  // the GRADIENT needs beyond them -- Y = L^-T and K^-1 = Y Y^T, 2/3 of an evaluation's flops -- runs on the fp32 tile
  // kernel (157 TFLOP/s) from a float copy of L, in float buffers of their own (api.hip: "mixed gradient")
  int grad_prec = 64;
  float *g32A = nullptr, *g32L = nullptr, *g32Y = nullptr, *g32D = nullptr;  // R / K^-1, copy of L, Y, copy of Dinv
  int64_t g32_cap = 0;
  int produce_panels = 4;                // 256-panels per super-panel of Produce's substitution
  int produce_small_below = 1024;        // Produce runs alone on the GPU: 64 x 64 tiles below this many 128-tiles (common.h: GemmGrid)
  int produce_tinv = 1;                  // option: 0 = Produce substitutes panel by panel (round 3)
  bool tinv_valid = false, tinv_pending = false;  // assembled for the current factor
  int64_t tinv_sig = 0;                  // super-panel layout it was assembled for
  // Produce for few test points (M <= produce_small_max, fp64, one GPU): ONE persistent launch that reads the factor
  // once (trsm_small.hip) instead of the tile-kernel chain; 0: off
  int produce_small_max = 64;
  void *small_ws = nullptr;    // its workspace (solution / w blocks, counters, partial sums)
  size_t small_ws_bytes = 0;
  int produce_groups = 2;      // Produce: independent substitution chains (streams) over the test points' tile rows
  int krag = 1;                // the inverse's updates skip the zero triangle of a super-panel of Y (common.h: GemmGrid::krag0)
  int ard_mfma_min = 1;        // ARD kernels (one radial term) with at least this many dimensions reduce the
                               // gradient on the matrix cores (grad_mfma.hip); 65: never
  int kinv_split = 60;         // percent of the columns whose part of K^-1 = Y Y^T is launched DURING the sweep (api.hip; 0: off)
  int64_t kinv_c1 = 0;         // ... columns that launch covered in the current factorisation (0: none)
  int kinv_fused = -1;          // ... and accumulates K^-1 = sum_P Y_P Y_P^T behind it, one rank-k update
                               // per super-panel of Y (0: one LAUUM launch over the finished Y in Gradient; -1: by size)
  int inv_prio = 0;            // 0: the inverse's streams at normal priority; 1: its bulk updates (s2) low;
                               // 2: bulk and chain (s2, st) low
  bool kinv_pending = false;   // K^-1 is being accumulated on sk (wait for EV_KINV)
  bool trtri_done = false;     // Y = L^-T of the current factor is (being) computed
  bool trtri_pending = false;  // ... and still running on st/s2 (wait for EV_TRTRI)
  bool ydone_valid = false;    // EV_YDONE was recorded by the factorisation trtri_pending refers to
  bool alpha_pending = false;   // alpha was enqueued on sp; consumers on s wait for ev_alpha
  // state
  std::vector<double> theta_s, theta_n;
  bool have_data = false, factored = false, have_alpha = false, have_kinv = false;
  bool observed = false, with_obs = false;
  double lml = 0.0;
  double yta = 0.0;      // y^T alpha of the last factorisation (fp32 path: of the refined alpha)
  int trace_fp64 = 1;    // fp32 path: tr(alpha alpha^T - K^-1) summed in fp64 from Y, scale component by its identity
  double cond_lb = 1.0;  // (max L_ii / min L_ii)^2 of the last factorisation
  double cond_limit = 1e16;  // gonum's mat.ConditionTolerance
  std::vector<double> grad_cache;
  bool grad_valid = false;
  int64_t notpd = -1;
  std::string err;
  GemmProfile prof;
  // HIP-event timing of the O(N^2) kernels (gogp_profile_read_aux): class -> event pairs
  std::vector<hipEvent_t> aux_ev[GOGP_PROF_NCLASS];
  size_t aux_used[GOGP_PROF_NCLASS] = {0, 0, 0, 0};
};

// Bracket the launches of one O(N^2) kernel class with events on their stream (only while
// profiling is enabled).
struct AuxTimer {
  gogp_handle *h;
  int cls;
  hipStream_t s;
  hipEvent_t e1 = nullptr;
  AuxTimer(gogp_handle *h_, int cls_, hipStream_t s_) : h(h_), cls(cls_), s(s_) {
    if (!h->prof.on || cls < 0 || cls >= GOGP_PROF_NCLASS) return;
    auto &pool = h->aux_ev[cls];
    size_t &used = h->aux_used[cls];
    while (pool.size() < used + 2) {
      hipEvent_t e = nullptr;
      (void)hipEventCreate(&e);
      pool.push_back(e);
    }
    (void)hipEventRecord(pool[used], s);
    e1 = pool[used + 1];
    used += 2;
  }
  ~AuxTimer() {
    if (e1) (void)hipEventRecord(e1, s);
  }
};

#define HIPCHK(h, call)                                                                \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      char buf_[512];                                                                  \
      snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
               __FILE__, __LINE__);                                                    \
      (h)->err = buf_;                                                                 \
      (void)hipGetLastError(); /* reset the sticky error: later calls must not see it */ \
      return (e_ == hipErrorOutOfMemory) ? GOGP_ENOMEM : GOGP_EHIP;                     \
    }                                                                                  \
  } while (0)

// every work stream of the handle (the communication stream of a sharded handle is dist2d's)
static inline std::array<hipStream_t, 6> work_streams(const gogp_handle *h) {
  return {h->s, h->sp, h->s2, h->st, h->sl, h->sk};
}

static inline int fail(gogp_handle *h, int code, const char *msg) {
  if (h) h->err = msg;
  return code;
}

// ---- cross-stream events -----------------------------------------------------------------
enum { EV_GRAM = 0, EV_FWD = 1, EV_ALPHA = 2, EV_INIT = 3, EV_TRTRI = 4, EV_W = 5, EV_ENTRY = 6, EV_KINV = 7,
       EV_YDONE = 8,  // Y is final (EV_TRTRI: everything on the inverse's chain stream is done, alpha = Y z included)
       EV_TINV = 9,   // T^-1 of every super-panel assembled (stream sk)
       EV_BASE = 10 };
// per panel p: EV_BASE + 4p + {0: panel p of L final, 1: next block column of A final,
//                              2: column panel p of Y final, 3: next column panel of R final}
static inline hipEvent_t ev(gogp_handle *h, size_t i) {
  while (h->evs.size() <= i) {
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    h->evs.push_back(e);
  }
  return h->evs[i];
}
// "a then b": work enqueued on `to` after this call waits for everything
// enqueued on `from` before it
static inline void order(gogp_handle *h, size_t i, hipStream_t from, hipStream_t to) {
  hipEvent_t e = ev(h, i);
  (void)gogp::rec_event_record(e, from);  // an explicit graph under construction takes these as dependencies (graphrec.h)
  (void)gogp::rec_stream_wait(to, e);
}


// ---- sharded evaluation (dist2d.hip) -----------------------------------------------------------
int gogp_upload_params(gogp_handle *h);            // api.hip: theta -> DevParams on h->s
void gogp_dist_destroy(gogp_handle *h);
int gogp_dist_sync(gogp_handle *h);                // drain the communication stream
int gogp_dist_ensure_n(gogp_handle *h, int64_t n);  // sizes + buffers of this rank's shard
int gogp_dist_factorize(gogp_handle *h, bool want_kinv);
int gogp_dist_gradient_sums(gogp_handle *h, double *hacc /* NACC, pinned host */);
int gogp_dist_produce(gogp_handle *h, const double *Z, int64_t m, double *mu, double *sigma);
int gogp_dist_get_factor(gogp_handle *h, double *Lout /* n*n, every rank */);
// rows != nullptr: nrows selected rows of L (nrows x n); rows == nullptr: the diagonal (n); collective
int gogp_dist_get_factor_part(gogp_handle *h, const int64_t *rows, int64_t nrows, double *out);
int gogp_dist_set_factor(gogp_handle *h, const double *Lin /* n*n, every rank */, const double *alpha /* n */);
