/*
 * gogp_hip.h -- C ABI of the MI355X-native GP-regression hot path.
 *
 * This is the drop-in boundary for infergo-ml/gogp's gp.GP hot path
 * (reference: gp/gp.go).  Every entry point names the reference interface it
 * replaces.  The library behind it (libgogp_hip.so) is hand-written HIP for
 * gfx950; there is no CPU fallback: if no HIP device is usable every compute
 * entry point returns GOGP_EHIP.
 *
 * Conventions
 *   - all functions are extern "C", return an int status (GOGP_OK == 0),
 *     take plain pointers and sizes; no C++/torch types cross the boundary;
 *   - host pointers are borrowed for the duration of the call only (cgo rule:
 *     no Go pointer is retained); device memory is owned by the handle;
 *   - a handle is NOT safe for concurrent use (same as a gp.GP value, whose
 *     methods mutate its fields: gp/gp.go:84,384-385); different handles may
 *     be used from different threads;
 *   - matrices are row-major doubles.
 */
#ifndef GOGP_HIP_H
#define GOGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------- */
#define GOGP_OK 0
#define GOGP_EARG 1   /* bad argument (wrong length, NULL, unknown kind)       */
#define GOGP_ENOTPD 2 /* K not positive definite: gp/gp.go:228-230 Factorize   */
#define GOGP_EHIP 3   /* HIP runtime error / no device / extension unusable    */
#define GOGP_ESTATE 4 /* call order (Gradient before Observe, ...)             */
#define GOGP_ENOMEM 5
#define GOGP_ECOND 6  /* K factored but numerically singular: gonum's Condition error
                         (cond > 1e16) from Cholesky.SolveVecTo / SolveTo, passed on by
                         gp/gp.go:233-236,338-340.  Here: (max L_ii/min L_ii)^2 > 1e16, a
                         lower bound of cond_2(K).  The factor, alpha and the LML are
                         still stored (as in gonum, which fills the result and returns the
                         error); the host shim decides: Absorb returns it as an error,
                         Observe panics, exactly where the reference does.          */

/* ---- kernel descriptors --------------------------------------------------
 * The reference accepts any Go value implementing
 *     type Kernel interface { Observe([]float64) float64; NTheta() int }
 * (gp/gp.go:14-17) and calls it once per pair (gp/gp.go:110-111).  A device
 * path needs a closed description instead: the similarity kernel is a SUM of
 * up to GOGP_MAX_TERMS terms, each  c * f(r)  with f one of the reference's
 * primitives (kernel/kernel.go).  This covers every kernel the reference
 * tree defines:
 *   kernel.Normal/Matern32/Matern52/Periodic       (kernel/kernel.go:23-92)
 *   c*Matern32                      (tutorial/barebones/kernel/kernel.go:14-16)
 *   c1*Matern52(l1)+c2*Periodic(l2,10p) (tutorial/hyperpriors/kernel/kernel.go:12-25)
 * For NDim > 1 (the reference primitives are 1-D, kernel/kernel.go:15-17) the
 * distance is  r^2 = sum_d ((xa_d-xb_d)/l_d)^2  (all l_d equal unless ard),
 * which reduces to the reference formulas at NDim == 1.
 */
#define GOGP_MAX_TERMS 4
#define GOGP_MAX_NDIM 64

enum gogp_simil_kind {
  GOGP_K_NORMAL = 0,            /* exp(-r^2/2)            kernel/kernel.go:23-26 */
  GOGP_K_MATERN32 = 1,          /* (1+s3 r)exp(-s3 r)     kernel/kernel.go:70-73 */
  GOGP_K_MATERN52 = 2,          /* (1+s5 r+1*r^2)exp(-s5 r): the reference's Go
                                   constant expression 5/3 is INTEGER division,
                                   i.e. 1 (kernel/kernel.go:91,
                                   kernel/ad/kernel.go:130)                      */
  GOGP_K_MATERN52_TEXTBOOK = 3, /* (1+s5 r+(5/3) r^2)exp(-s5 r)                  */
  GOGP_K_PERIODIC = 4           /* exp(-2 d^2), d=sin(pi|dx|/p)/l
                                                          kernel/kernel.go:44-47 */
};

enum gogp_noise_kind {
  GOGP_NOISE_CONSTANT = 0, /* kernel.ConstantNoise(std): var=std^2, NTheta=0
                              kernel/noise.go:21-34                             */
  GOGP_NOISE_UNIFORM = 1,  /* scale*kernel.UniformNoise: var=scale*std^2,
                              NTheta=1  kernel/noise.go:39-53; scale as in
                              tutorial/barebones/kernel/kernel.go:25-31         */
  GOGP_NOISE_CONSTANT_PARAM = 2 /* constant variance noise_std^2 WITH one parameter
                              the Gram matrix does not depend on (NTheta=1, its
                              gradient component is 0): the noise kernel of
                              tutorial/anynoise/kernel/kernel.go:26-35, whose
                              parameter only feeds the priors                   */
};

typedef struct gogp_term {
  int32_t kind;       /* enum gogp_simil_kind                                   */
  int32_t scale_idx;  /* index in ThetaSimil of the output scale c; -1: c == 1  */
  int32_t len_idx;    /* index in ThetaSimil of the (first) length scale        */
  int32_t ard;        /* 0: one length scale; 1: ndim consecutive length scales */
  int32_t period_idx; /* GOGP_K_PERIODIC: index of the period parameter         */
  int32_t reserved;
  double period_mult; /* effective period = period_mult * theta[period_idx]
                         (the 10*x[p] of tutorial/hyperpriors/kernel/kernel.go:24) */
} gogp_term;

typedef struct gogp_desc {
  int32_t ndim;         /* gp.GP.NDim                      gp/gp.go:22          */
  int32_t nterms;       /* 1..GOGP_MAX_TERMS                                    */
  int32_t ntheta_simil; /* gp.GP.Simil.NTheta()                                 */
  int32_t noise_kind;   /* enum gogp_noise_kind; gp.GP.Noise; nil => CONSTANT
                           with std 1e-5 (gp/gp.go:43-48)                       */
  double noise_std;     /* CONSTANT: the std                                    */
  double noise_scale;   /* UNIFORM: variance = noise_scale * std^2              */
  gogp_term terms[GOGP_MAX_TERMS];
} gogp_desc;

typedef struct gogp_handle gogp_handle;

/* ---- lifecycle ------------------------------------------------------------ */

/* Validate a descriptor; returns GOGP_OK or GOGP_EARG.  Pure host code. */
int gogp_desc_check(const gogp_desc *desc);

/* Number of noise parameters, Noise.NTheta(): 0 (CONSTANT) or 1. */
int gogp_desc_ntheta_noise(const gogp_desc *desc);

/* Create a handle on HIP device `device` (-1: the current device).
 * Replaces: constructing a gp.GP{NDim,Simil,Noise} value (gp/gp.go:20-24). */
int gogp_create(const gogp_desc *desc, int device, gogp_handle **out);
void gogp_destroy(gogp_handle *h);

/* Last error text for this handle (never NULL). With h == NULL: the text of
 * the last failed gogp_create on this thread. */
const char *gogp_last_error(const gogp_handle *h);

/* After GOGP_ENOTPD: 0-based index of the failing pivot, else -1. */
int64_t gogp_notpd_index(const gogp_handle *h);

/* ---- data ------------------------------------------------------------------
 * Replaces: assigning gp.GP.X ([][]float64, n slices of NDim) and gp.GP.Y
 * (gp/gp.go:27-28,84).  X is packed row-major n x ndim by the caller's shim.
 * The data are copied to the device; n == 0 is legal (gp/gp.go:101-104). */
int gogp_set_data(gogp_handle *h, const double *X, const double *y, int64_t n);

/* Same, but X and y already live in device memory of the handle's device
 * (used by bench.py so the timed region starts with inputs resident in HBM). */
int gogp_set_data_device(gogp_handle *h, const double *dX, const double *dy,
                         int64_t n);

/* ---- the hot path ----------------------------------------------------------*/

/* gp.GP.Absorb (gp/gp.go:80-87) minus the data assignment (gogp_set_data):
 * Gram build (gp/gp.go:109-156,220-225), Cholesky (gp/gp.go:228), alpha
 * (gp/gp.go:232-236), WITHOUT gradient.  theta_* are natural-scale
 * ThetaSimil/ThetaNoise (gp/gp_test.go:29).  Non-PD => GOGP_ENOTPD. */
int gogp_absorb(gogp_handle *h, const double *theta_simil,
                const double *theta_noise);

/* gp.GP.Observe, hyperparameters-only form (gp/gp.go:370-373,374-413):
 * x = log-transformed [ThetaSimil | ThetaNoise] (len P); X, Y as set by
 * gogp_set_data.  Computes exp(x), Gram, Cholesky, alpha, and returns
 * LML (gp/gp.go:244-253) in *lml.  Unlike the reference x is NOT mutated
 * (the reference exp()s and log()s it back in place, gp/gp.go:378-381,
 * 408-410).  A following gogp_gradient returns d LML / d x.
 * Errors are returned as codes; the Go shim turns them into the reference's
 * panic (gp/gp.go:402-405). */
int gogp_observe(gogp_handle *h, const double *x, int64_t len, double *lml);

/* gp.GP.Observe, full form (gp/gp.go:366-369,386-397): x = [log theta (P) |
 * x_0..x_{n-1} (ndim each) | y_0..y_{n-1}], n = (len-P)/(ndim+1); replaces the
 * handle's data by the inputs/outputs carried in x.  len must equal
 * P + n*(ndim+1) for an integer n (reference: panic("len(x)"), gp/gp.go:398-400). */
int gogp_observe_full(gogp_handle *h, const double *x, int64_t len, double *lml);

/* gp.GP.LML (gp/gp.go:244-253) of the last absorb/observe; 0 when n == 0. */
int gogp_lml(gogp_handle *h, double *lml);

/* gp.GP.Gradient (gp/gp.go:418-499): gradient of LML w.r.t. the argument of
 * the last gogp_observe[_full]: P entries d/d log theta, then -- after
 * gogp_observe_full -- n*ndim entries d/d x_i,d and n entries d/d y_i = -alpha_i
 * (gp/gp.go:488-493).  `len` must be that length.  n == 0 => zeros
 * (gp/gp.go:427-430). */
int gogp_gradient(gogp_handle *h, double *grad, int64_t len);

/* k independent Observe + Gradient evaluations at once: handle hs[i] (each created and given
 * its data separately; they may hold the same data) evaluates x[i*len .. (i+1)*len) from its own
 * host thread, so the dependent launch chains of the k evaluations overlap on the GPU.
 * Counterpart: the reference's optimiser evaluating candidates concurrently
 * (optimize.Settings.Concurrent = NTASKS, tutorial/tutorial.go:30,141).  status (may be NULL)
 * receives the k return codes; the result is the first non-zero one. */
int gogp_observe_gradient_batch(gogp_handle **hs, int k, const double *x, int64_t len,
                                double *lmls /* k */, double *grads /* k*len */,
                                int *status /* k */);

/* k candidate parameter vectors on ONE handle's data, evaluated in one launch sequence: x is
 * k x len row-major (len = P: log theta only), lmls[c] and grads[c*len ..] receive what
 * gogp_observe + gogp_gradient would return for candidate c.  Every kernel of the sweep is launched
 * once for all candidates (candidate index on the grid's z axis), so the dependent chain of small
 * launches that bounds one evaluation below N ~ 8192 is paid once per batch.  Same counterpart as
 * above (concurrently evaluated candidates of the reference's optimiser: a line search's trial
 * points, multi-start restarts).  The handle's own state -- data, the factorisation of an earlier
 * Absorb / Observe, Produce -- is not touched; the candidates live in k arena slots of
 * ~3 x 8 N^2 bytes each that stay allocated until the handle is destroyed.  status[c] (may be
 * NULL): GOGP_OK, GOGP_ENOTPD (lmls[c] = NaN, gradient zeros), GOGP_ECOND (values still
 * returned) or GOGP_EARG (parameters not finite); the result is the first non-zero one.
 * k <= GOGP_MAX_CANDIDATES.  fp64 handle on one GPU: as described.  precision = 32: the candidates pass through
 * ONE arena slot one after the other (same results and the same untouched handle; the float kernels carry no
 * candidate index).  Sharded handle (collective: every rank with the same candidates): evaluated one after the
 * other in the shards' own tiles, and the handle afterwards holds the factorisation of the last candidate that was
 * factorised (one refused with GOGP_EARG is skipped; after a last candidate that is not positive definite the handle
 * holds no factorisation).  The per-candidate contract is the same on every kind of handle: all k slots of lmls /
 * grads / status are written, GOGP_EARG and GOGP_ENOTPD mark their own candidate only; only a transport or HIP
 * failure ends the call early (status of the candidates not reached: GOGP_ESTATE). */
#define GOGP_MAX_CANDIDATES 16
int gogp_observe_gradient_candidates(gogp_handle *h, int k, const double *x, int64_t len,
                                     double *lmls /* k */, double *grads /* k*len */,
                                     int *status /* k */);

/* Diagnostics of the candidates' launch graph (option "graph"): nodes of the graph in use (0: none yet, or the
 * stream path), whether the runtime refused an explicitly built graph (then the stream path is used for good).
 * No reference counterpart. */
int gogp_graph_info(const gogp_handle *h, int64_t *nodes, int *refused);

/* gp.GP.Produce (gp/gp.go:258-360): predictive mean and standard deviation of
 * the latent function at m points Z (row-major m x ndim).  sigma_j =
 * sqrt(k(z_j,z_j) - (Kstar^T K^-1 Kstar)_jj), unclamped like the reference
 * (gp/gp.go:356: a rounding-negative argument yields NaN).  With no
 * observations: mu = 0, sigma = sqrt(prior) (gp/gp.go:343-347). */
int gogp_produce(gogp_handle *h, const double *Z, int64_t m, double *mu,
                 double *sigma);

/* ---- cached state (gp.GP.L, gp.GP.Alpha: gp/gp.go:35-36,255-257) ---------- */
int64_t gogp_n(const gogp_handle *h);
int gogp_get_alpha(gogp_handle *h, double *alpha /* n */);
/* Lower Cholesky factor, row-major n x n, upper triangle zero-filled.
 * (gonum stores U = L^T; the Go shim transposes when filling gp.GP.L.) */
int gogp_get_factor(gogp_handle *h, double *L /* n*n */);
/* Selected rows of the factor: out is nrows x n row-major, row r = L[rows[r], 0..n-1]
 * (zeros right of the diagonal); and its diagonal (n doubles).  gp.GP.L at sizes where
 * the whole n x n matrix is not wanted on the host (2*sum(log diag) is the LogDet of
 * gp/gp.go:250). */
int gogp_get_factor_rows(gogp_handle *h, const int64_t *rows, int64_t nrows,
                         double *out /* nrows*n */);
int gogp_get_factor_diag(gogp_handle *h, double *diag /* n */);
/* Restore stored results so that gogp_produce works without re-absorbing
 * ("Produce on stored results", gp/gp.go:255-257). */
int gogp_set_factor(gogp_handle *h, const double *theta_simil,
                    const double *theta_noise, const double *L /* n*n */,
                    const double *alpha /* n */);

/* ---- one evaluation sharded over several GPUs: 2-D block-cyclic ---------------------
 * One process per GPU.  The ranks form a Pr x Pc process grid (rank = pr*Pc + pc; Pr must
 * divide Pc: 1x1, 1x2, 2x2, 2x4 for 1/2/4/8 GPUs, gogp_dist_grid).  The Gram matrix is cut into
 * 512x512 tiles; tile (I,J) lives on the GPU at grid position (I mod Pr, J mod Pc), and every
 * rank allocates ONLY its own tiles (of K, of the factor L and of Y = L^-T): memory per rank
 * is 1/(Pr*Pc) of the single-GPU footprint (gogp_dist_local_bytes).  X, y (a few MB) are
 * replicated, so every rank builds its own tiles of K with no communication.  Per block
 * column P of the blocked right-looking Cholesky: the owner of the diagonal tile factors and
 * inverts it and sends the inverse out; the process column that owns block column P solves
 * its tiles of the panel; the panel is then sent along the process rows (each rank gets the
 * tiles of its own tile rows) and, transposed, along the process columns (the tiles of its
 * own tile columns); every rank updates its trailing tiles on MFMA.  The triangular inverse
 * Y = L^-T and K^-1 = Y Y^T run right behind on the same layout with the same exchange, the
 * gradient reduction runs on the local tiles of K^-1, and the partial sums meet in one
 * all-reduce.  All ranks call the SAME sequence of gogp_set_data / gogp_absorb / gogp_observe
 * / gogp_gradient / gogp_produce collectively and get the same LML, gradient, alpha, mu, sigma.
 * The reference has no counterpart (single process, goroutines only: gp/gp.go:165-213).
 *
 * Transport, chosen at initialisation (call right after gogp_create, before gogp_set_data):
 *   gogp_dist_init_rccl       RCCL over xGMI from inside the library: grouped ncclSend /
 *                             ncclRecv between peers and ncclAllReduce, enqueued on a
 *                             communication stream and ordered against the compute streams
 *                             by events (no host synchronisation per panel).  The host layer
 *                             only has to hand every rank the same 128-byte unique id
 *                             (made on one rank by gogp_dist_unique_id).
 *   gogp_dist_init_callbacks  host-synchronous exchange through two callbacks over HOST
 *                             buffers (the library stages payloads through pinned memory):
 *                               exchange(user, ops, nops): perform all transfers of the list
 *                                 (is_send: send `bytes` from buf to rank `peer`; else receive
 *                                 into buf); returns after all have completed.  Every pair of
 *                                 ranks lists its mutual transfers in the same order.
 *                               allreduce(user, host_buf, count): sum doubles over the ranks.
 *                             For rehearsals where RCCL cannot run (several ranks sharing one
 *                             GPU, gloo) and for hosts with their own transport (MPI, sockets). */
#define GOGP_UNIQUE_ID_BYTES 128
typedef struct gogp_xfer {
  int32_t peer;    /* rank of the other side                                       */
  int32_t is_send; /* 1: send from buf, 0: receive into buf                        */
  void *buf;       /* host memory, valid until the callback returns                */
  int64_t bytes;
} gogp_xfer;
typedef int (*gogp_exchange_fn)(void *user, const gogp_xfer *ops, int32_t nops);
typedef int (*gogp_allreduce_fn)(void *user, double *host_buf, int64_t count);
/* Default process grid for `nranks` GPUs: 1x1, 1x2, 2x2, 2x4, (16: 4x4). */
int gogp_dist_grid(int nranks, int *prow, int *pcol);
int gogp_dist_unique_id(void *id128 /* GOGP_UNIQUE_ID_BYTES */);
int gogp_dist_init_rccl(gogp_handle *h, int rank, int nranks, int prow, int pcol,
                        const void *id128);
int gogp_dist_init_callbacks(gogp_handle *h, int rank, int nranks, int prow, int pcol,
                             gogp_exchange_fn exchange, gogp_allreduce_fn allreduce,
                             void *user);
/* Ranks of the communicator as the transport itself counts them (RCCL: ncclCommCount; callbacks:
 * nranks), *is_rccl (may be NULL) = 1 for the RCCL transport; -1 on an unsharded handle. */
int gogp_dist_comm_ranks(const gogp_handle *h, int *is_rccl);
/* Pre-flight of the transport, collective: phase 0 = ONE group with a send of `count` doubles to rank+1
 * and a receive from rank-1 (the shape of every panel exchange of the sweep), phase 1 = one all-reduce of
 * `count` doubles; payloads are checked.  Returns after the communication stream has drained, so a
 * transport that hangs shows up as a call that does not return (bench.py runs both under a watchdog
 * before the first sharded evaluation).  No reference counterpart. */
int gogp_dist_selftest(gogp_handle *h, int phase, int64_t count);
/* Device bytes this rank holds for the N-dependent state (its tiles of K / L / Y, panel
 * buffers, block inverses); 0 before gogp_set_data or on an unsharded handle. */
int64_t gogp_dist_local_bytes(const gogp_handle *h);

/* ---- measurement hooks (bench.py / tests; not part of the reference API) --- */

/* Enable (1) / disable (0) HIP-event timing of every launch of the dominant
 * kernel family (the fp64 MFMA GEMM/SYRK tile kernel) on the stream it is
 * launched on. */
int gogp_profile_enable(gogp_handle *h, int on);
/* Since the last reset: sum of the event-measured launch durations (ms), number
 * of launches, flops launched, and the length (ms) of the union of the launch
 * intervals (launches overlap: the hot path runs on several streams).  Resets
 * the accumulators. */
int gogp_profile_read(gogp_handle *h, double *gemm_ms, int64_t *gemm_launches,
                      double *gemm_flops, double *gemm_busy_ms);

/* Every launch of that kernel family since gogp_profile_enable(h, 1), in launch order: start and end in
 * ms since the first launch started, flops launched, tag = mode * 1e8 + (K / 16) * 1e5 + tiles (mode 0
 * rectangular, 1 lower/SYRK, 2 LAUUM).  *n = number of launches (may exceed cap; cap entries are
 * written).  The timeline behind roofline.achieved; tools/launch_timeline.py bins it. */
int gogp_profile_read_launches(gogp_handle *h, int64_t cap, double *t0_ms, double *t1_ms, double *flops,
                               int64_t *tag, int64_t *n);

/* The same for the bandwidth-bound O(N^2) kernels: sum of the event-measured durations
 * (ms) and number of timed launch groups of one class since the last read. */
#define GOGP_PROF_GRAM 0  /* Gram build (the main-stream part: all but the first 512 columns) */
#define GOGP_PROF_GRAD 1  /* fused gradient reduction over K^-1 */
#define GOGP_PROF_CROSS 2 /* cross-covariance build of Produce */
#define GOGP_PROF_NCLASS 4
int gogp_profile_read_aux(gogp_handle *h, int cls, double *ms, int64_t *launches);

/* Scheduling knobs (the results do not depend on them beyond rounding; the defaults are the
 * measured optimum).  Unknown name or out-of-range value: GOGP_EARG.
 *   "lookahead"    1 | 0   panel chain on its own high-priority stream / everything in order
 *                          on one stream                                           (default 1)
 *   "eager"        1 | 0   Observe also runs the triangular inverse (gradient preparation)
 *                          behind the Cholesky sweep / Gradient computes it lazily  (default 1)
 *   "superpanel"   1..8    256-wide panels per trailing update (K = 256 * value)    (default 2)
 *   "precision"    64 | 32 fp64 throughout / the N x N matrices and the O(N^3) products in fp32
 *                          (v_mfma_f32_32x32x2_f32) with fp64 inputs, kernel evaluation, diagonal
 *                          blocks, vectors and reductions: BASELINE configs[4].  Set it before
 *                          gogp_set_data (it re-sizes the buffers) and before gogp_dist_init_*
 *                          (a shard then holds float tiles and exchanges float panels)  (default 64)
 *   "refine_steps" 0..8    precision 32 only: steps of iterative refinement of alpha against
 *                          the exact (fp64, recomputed) Gram matrix                    (default 1)
 *   "cond_limit_log10" 1..300  GOGP_ECOND threshold 10^value -- gonum's package variable
 *                          mat.ConditionTolerance                                   (default 16)
 *   "superpanel_head" -1..8, "head_remaining" >= 0   wider super-panels while more than head_remaining
 *                          panels are still to come (the chain has slack there; bulk updates with a longer
 *                          K are more efficient); 0 switches it off; -1: 3, on the fp32 path 4
 *                                                                                         (default -1, 16)
 *   "kinv_fused"   -1|0|1  K^-1 accumulated behind the triangular inverse as one rank-k update per
 *                          super-panel of Y (1) or formed by one launch over the finished Y in
 *                          gogp_gradient (0); -1: fused up to N = 10240                    (default -1)
 *   "kinv_split"   0..95   where K^-1 is not fused (fp64, N > 10240): once this percentage of the columns of Y is
 *                          final, their part of K^-1 = Y Y^T is one launch inside the sweep and gogp_gradient's launch
 *                          adds the rest (same sums, same order: bit-identical); 0: one launch       (default 60)
 *   "chain_prio"   -1..2   the chains' tile-kernel launches raise their waves' issue priority: 0 never, 1 the
 *                          skinny (64x64-tile) ones, 2 all of them, -1 the skinny ones up to N = 6144  (default -1)
 *   "ktri"         1 | 0   panel solves skip the zero half of the (lower triangular) block inverse (default 1)
 *   "ard_mfma_min_dims" 1..65  ARD kernels with one radial term and at least this many dimensions run
 *                          the gradient reduction with distances and per-dimension sums on the matrix
 *                          cores (grad_mfma.hip); 65: never                                 (default 1)
 *   "graph"        0..2    gogp_observe_gradient_candidates: on its second identical use the launch sequence
 *                          becomes a hipGraph and is replayed (parameters and data may change, sizes may not).
 *                          1: round 2's linear graph from stream capture, N <= 1024 (larger sizes: streams);
 *                          2: an explicitly built graph -- one node per launch / copy, the sweep's real
 *                          cross-stream dependencies as edges, no stream capture -- up to N = 8192:
 *                          bit-identical to the stream path and, on this runtime, slower than it (DESIGN.md
 *                          section 4); 3: the same recorder as ONE chain in enqueue order (diagnostics);
 *                          0: streams                                                       (default 1)
 *   "produce_tinv", "produce_panels", "produce_groups", "produce_small_below"
 *                          gogp_produce: whole super-panels of `produce_panels` 256-column panels solved through
 *                          the inverse of the factor's diagonal block (1) or panel by panel (0); the test
 *                          points' tile rows on `produce_groups` independent chains; 64 x 64 tiles for launches
 *                          below that many 128-tiles                                  (default 1, 4, 2, 1024)
 *   "gradient_precision" 64 | 32   on an fp64 handle: 32 runs what only the gradient needs -- Y = L^-T and
 *                          K^-1 = Y Y^T, 2/3 of an evaluation's flops -- on the fp32 tile kernel from a float copy of
 *                          the fp64 factor; factorisation, LML, alpha and Produce stay fp64 bit for bit (default 64).
 *                          Offered for kernels of ONE term with an output scale (GOGP_EARG otherwise): there the
 *                          cancelling components come from closed forms and the gradient stays ~1e-8 from the fp64
 *                          one (1e-6 from the oracle in the tests); a sum of terms would read its scale components
 *                          off the float K^-1 (1.9e-4 measured), beyond the reference's own 1e-4 (gp_test.go:170,248)
 *   "trace_fp64"   1 | 0   float K^-1 (precision = 32, gradient_precision = 32): tr(alpha alpha^T - K^-1) summed in
 *                          fp64 from Y and the output-scale component from its closed form          (default 1)
 *   "diag_fp64"    1 | 0   fp32 path (single GPU and float shards): the diagonal blocks' / tiles' trailing updates are
 *                          summed in fp64 from the float panels (diagsyrk.hip) and the fp64 diagonal-block kernel
 *                          factors THAT image; 0: it widens the float matrix's block (rounds 2-4)    (default 1)
 *   "krag"         1 | 0   the triangular inverse's updates skip the zero triangle of a super-panel of Y (default 1)
 *   "chain_split"  -1 | 0 | 1 | 2   fp64, the factorisation's dependency chain per 256-panel: 0 one workgroup factors and
 *                          inverts the 256 x 256 diagonal block, the panel solve is a K = 256 product with that inverse;
 *                          1 two 128-halves (factor + inverse each) with the products between them on the tile kernel
 *                          and X10 of the block inverse formed off the chain; 2 per 128 columns ONE launch factors the
 *                          diagonal 128-block and forward-substitutes every panel row on the way (panel128.hip), the
 *                          block inverses are formed off the chain from the finished factor; -1: 2 where the
 *                          evaluation is latency-bound (N <= 8192) or no fp64 inverse runs beside the factorisation
 *                          (Absorb, eager = 0); above that, beside the inverse, 0 -- and 2 for the super-panels with at
 *                          most "chain_tail" rows left                                                (default -1)
 *   "tiny"         1 | 0   fp64, one GPU, N <= 128 observations (the reference's own case studies): Gram matrix, factor, block
 *                          inverse, z, alpha and K^-1 in ONE launch of one workgroup instead of the general sweep's ~15
 *                          dependent launches; 0: the general sweep                                   (default 1)
 *   "chain_slabs"  0..8    chain_split = 2: 64-row slabs of the panel per workgroup of the chain step; the result does
 *                          not depend on it; 0: one while the launch has at most a workgroup per compute unit, up to 4
 *                          beyond (measured at N = 16384: 1 / 2 / 4 slabs 30.1 / 32.0 / 35.4 ms Observe only) (default 0)
 *   "chain_tail"   0..2^20 rows (see chain_split; measured slower at N = 16384 and 32768)            (default 0)
 *   "produce_small_max" 0..64   gogp_produce with up to this many test points (fp32 path: up to 16 of them): ONE
 *                          persistent launch that reads the factor once (trsm_small.hip; float factors are widened in
 *                          registers, the sums are fp64) instead of the tile-kernel chain; 0: never  (default 64)
 * No reference counterpart (gp.GP.Parallel, gp/gp.go:30-31, only switches goroutines on). */
int gogp_set_option(gogp_handle *h, const char *name, int64_t value);

/* Library build info: "gogp_hip <version> gfx950 ..." */
const char *gogp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GOGP_HIP_H */
