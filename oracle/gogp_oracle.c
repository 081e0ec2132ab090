/*
 * gogp_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp64)
 * of the reference algorithm of infergo-ml/gogp's GP hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, link or call this file.  The product (gogp_amd/, libgogp_hip.so)
 * never does; it has no CPU fallback.
 *
 * What is restated, with the reference lines each function follows:
 *   kernel formulas          kernel/kernel.go:23-26,44-47,70-73,89-92
 *   noise kernels            kernel/noise.go:27-30,47-49
 *   GP.defaults              gp/gp.go:45-57   (done by the caller: descriptor)
 *   GP.absorb  (Gram, dK)    gp/gp.go:89-239
 *   GP.LML                   gp/gp.go:244-253
 *   GP.Produce               gp/gp.go:258-360
 *   GP.Observe               gp/gp.go:374-413
 *   GP.Gradient              gp/gp.go:418-499
 *
 * Third-party arithmetic that is NOT in /root/reference and is restated from
 * its published algorithm:
 *   gonum.org/v1/gonum v0.9.3 (go.mod:7): mat.Cholesky.Factorize (LAPACK
 *   Dpotrf: K = U^T U), SolveVecTo / SolveTo (two triangular solves),
 *   LogDet (2 sum log U_ii), Mul, MulVec, Sub, Trace, Dot.  Restated here as
 *   an unblocked lower Cholesky (L = U^T), forward/back substitution and plain
 *   triple loops.
 *   bitbucket.org/dtolpin/infergo v1.2.2 (go.mod:6): reverse-mode AD of each
 *   pair evaluation (model.Gradient, gp/gp.go:113,137).  Restated as the
 *   closed-form partial derivatives of the same expressions w.r.t. the same
 *   argument vector [theta | xa | xb].
 *
 * Pinning: tests/test_oracle_golden.py checks this file against every known
 * answer of the reference's own tests (gp/gp_test.go:23-120 Produce cases,
 * gp/gp_test.go:180-229 LML cases, gp/gp_test.go:242-252 finite-difference
 * gradient check).  Those pin kernel.Normal with Constant/Uniform noise at
 * N <= 2, NDim = 1.  Matern32/52, Periodic, NDim > 1 and large N have no
 * reference vectors: for those the parity is "unpinned by the reference" and
 * rests on this restatement.
 *
 * The "faithful" entry points follow the reference operation by operation
 * (pair loop j >= i, dense dK per parameter, 1/2 tr(aa^T dK - K^-1 dK)).
 * They are O(P N^3) and meant for N up to a few hundred.
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include "../include/gogp_hip.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SQRT3 1.7320508075688772 /* kernel/kernel.go:51 */
#define SQRT5 2.2360679774997900 /* kernel/kernel.go:52 */

typedef struct gogp_oracle {
  gogp_desc desc;
  int64_t n;       /* len(gp.X) */
  int with_obs;    /* gp.withObs, gp/gp.go:32,386 */
  int has_dk;      /* dK built by the last absorb(withGradient) */
  double *theta_s; /* gp.ThetaSimil (natural scale) */
  double *theta_n; /* gp.ThetaNoise */
  double *X;       /* n x ndim */
  double *Y;       /* n */
  double *K;       /* n x n Gram matrix (kept for inspection) */
  double *L;       /* n x n lower Cholesky factor (= gonum's U^T) */
  double *alpha;   /* gp.Alpha */
  double **dK;     /* gp.dK: ndk dense n x n matrices */
  int64_t ndk;
} gogp_oracle;

static int ntheta_noise(const gogp_desc *d) {
  return (d->noise_kind == GOGP_NOISE_UNIFORM || d->noise_kind == GOGP_NOISE_CONSTANT_PARAM) ? 1 : 0;
}

/* ---- similarity kernel: value and gradient w.r.t. [theta | xa | xb] -------
 * The value follows kernel/kernel.go term by term; the gradient is what
 * infergo's tape (kernel/ad/kernel.go) yields for the same expression.
 * grad may be NULL (model.DropGradient, gp/gp.go:131). */
double gogp_oracle_simil(const gogp_desc *d, const double *theta, const double *xa,
                         const double *xb, double *grad) {
  int D = d->ndim, nt = d->ntheta_simil;
  double k = 0.0;
  if (grad)
    for (int i = 0; i < nt + 2 * D; i++) grad[i] = 0.0;
  for (int t = 0; t < d->nterms; t++) {
    const gogp_term *T = &d->terms[t];
    double c = T->scale_idx >= 0 ? theta[T->scale_idx] : 1.0;
    double f = 0.0;
    if (T->kind == GOGP_K_PERIODIC) {
      /* kernel/kernel.go:44-47: d := sin(pi*|xa-xb|/p)/l; exp(-2 d d) */
      double p = T->period_mult * theta[T->period_idx];
      double s2 = 0.0;
      for (int j = 0; j < D; j++) {
        double l = theta[T->len_idx + (T->ard ? j : 0)];
        double dd = sin(M_PI * fabs(xa[j] - xb[j]) / p) / l;
        s2 += dd * dd;
      }
      f = exp(-2.0 * s2);
      k += c * f;
      if (grad) {
        if (T->scale_idx >= 0) grad[T->scale_idx] += f;
        for (int j = 0; j < D; j++) {
          int li = T->len_idx + (T->ard ? j : 0);
          double l = theta[li];
          double dx = xa[j] - xb[j];
          double adx = fabs(dx);
          double phi = M_PI * adx / p;
          double dd = sin(phi) / l;
          /* df/dd = -4 d f */
          double dfdd = -4.0 * dd * f;
          grad[li] += c * dfdd * (-dd / l);
          /* d dd / d p_eff = cos(phi) * (-pi |dx| / p^2) / l ; p_eff = mult*theta */
          grad[T->period_idx] +=
              c * dfdd * (cos(phi) * (-M_PI * adx / (p * p)) / l) * T->period_mult;
          /* d|dx|/dxa = sign(dx) (0 at dx == 0, as math.Abs' tape derivative
           * is irrelevant there: the factor sin(phi) vanishes) */
          double sg = dx > 0 ? 1.0 : (dx < 0 ? -1.0 : 0.0);
          double ddx = cos(phi) * (M_PI / p) / l * sg;
          grad[nt + j] += c * dfdd * ddx;
          grad[nt + D + j] -= c * dfdd * ddx;
        }
      }
      continue;
    }
    /* radial kernels: r^2 = sum_j ((xa_j - xb_j)/l_j)^2 */
    double r2 = 0.0;
    for (int j = 0; j < D; j++) {
      double l = theta[T->len_idx + (T->ard ? j : 0)];
      double u = (xa[j] - xb[j]) / l;
      r2 += u * u;
    }
    double dfdr2 = 0.0; /* df / d(r^2) */
    switch (T->kind) {
    case GOGP_K_NORMAL: /* kernel/kernel.go:23-26 */
      f = exp(-r2 / 2);
      dfdr2 = -0.5 * f;
      break;
    case GOGP_K_MATERN32: { /* kernel/kernel.go:70-73 */
      double r = sqrt(r2);
      double e = exp(-SQRT3 * r);
      f = (1 + SQRT3 * r) * e;
      /* df/dr = -3 r e  =>  df/dr2 = df/dr / (2r) = -1.5 e */
      dfdr2 = -1.5 * e;
      break;
    }
    case GOGP_K_MATERN52: { /* kernel/kernel.go:89-92, 5/3 == 1 in Go */
      double r = sqrt(r2);
      double e = exp(-SQRT5 * r);
      f = (1 + SQRT5 * r + 1 * r * r) * e;
      /* df/dr = -(3 r + s5 r^2) e => df/dr2 = -(3 + s5 r)/2 e */
      dfdr2 = -0.5 * (3 + SQRT5 * r) * e;
      break;
    }
    case GOGP_K_MATERN52_TEXTBOOK: {
      double r = sqrt(r2);
      double e = exp(-SQRT5 * r);
      f = (1 + SQRT5 * r + (5.0 / 3.0) * r * r) * e;
      /* df/dr = -(5/3) r (1 + s5 r) e => df/dr2 = -(5/6)(1 + s5 r) e */
      dfdr2 = -(5.0 / 6.0) * (1 + SQRT5 * r) * e;
      break;
    }
    default:
      return NAN;
    }
    k += c * f;
    if (grad) {
      if (T->scale_idx >= 0) grad[T->scale_idx] += f;
      for (int j = 0; j < D; j++) {
        int li = T->len_idx + (T->ard ? j : 0);
        double l = theta[li];
        double dx = xa[j] - xb[j];
        double u = dx / l;
        /* d r2/d l_j = -2 u^2 / l ; d r2/d xa_j = 2 u / l */
        grad[li] += c * dfdr2 * (-2.0 * u * u / l);
        grad[nt + j] += c * dfdr2 * (2.0 * u / l);
        grad[nt + D + j] -= c * dfdr2 * (2.0 * u / l);
      }
    }
  }
  return k;
}

/* ---- noise kernel: value and gradient w.r.t. [theta_n | x] ---------------- */
double gogp_oracle_noise(const gogp_desc *d, const double *theta_n, const double *x,
                         double *grad) {
  (void)x;
  int nn = ntheta_noise(d);
  if (grad)
    for (int i = 0; i < nn + d->ndim; i++) grad[i] = 0.0;
  if (d->noise_kind == GOGP_NOISE_CONSTANT || d->noise_kind == GOGP_NOISE_CONSTANT_PARAM) {
    /* kernel/noise.go:27-30; tutorial/anynoise/kernel/kernel.go:31-33: a constant whatever
     * the (single, unused) parameter is -- the tape gradient of a constant is 0 */
    return d->noise_std * d->noise_std;
  }
  /* kernel/noise.go:47-49 times the tutorial's constant factor */
  double std = theta_n[0];
  if (grad) grad[0] = d->noise_scale * 2.0 * std;
  return d->noise_scale * std * std;
}

/* ---- lifecycle ------------------------------------------------------------ */
gogp_oracle *gogp_oracle_new(const gogp_desc *d) {
  gogp_oracle *o = (gogp_oracle *)calloc(1, sizeof(*o));
  if (!o) return NULL;
  o->desc = *d;
  /* gp/gp.go:50-56: zero theta vectors of the right length */
  o->theta_s = (double *)calloc((size_t)(d->ntheta_simil > 0 ? d->ntheta_simil : 1),
                                sizeof(double));
  o->theta_n = (double *)calloc(2, sizeof(double));
  return o;
}

static void free_dk(gogp_oracle *o) {
  if (o->dK) {
    for (int64_t i = 0; i < o->ndk; i++) free(o->dK[i]);
    free(o->dK);
  }
  o->dK = NULL;
  o->ndk = 0;
  o->has_dk = 0;
}

static void free_state(gogp_oracle *o) {
  free(o->X);
  free(o->Y);
  free(o->K);
  free(o->L);
  free(o->alpha);
  o->X = o->Y = o->K = o->L = o->alpha = NULL;
  free_dk(o);
}

void gogp_oracle_free(gogp_oracle *o) {
  if (!o) return;
  free_state(o);
  free(o->theta_s);
  free(o->theta_n);
  free(o);
}

/* ---- dense helpers (gonum restated) ---------------------------------------- */

/* Cholesky, lower, unblocked; returns 0 or 1+index of the failing pivot.
 * gonum: mat.Cholesky.Factorize -> lapack Dpotrf; false when not PD. */
static int64_t chol_lower(double *A, int64_t n) {
  for (int64_t j = 0; j < n; j++) {
    double *Aj = A + j * n;
    double s = Aj[j];
    for (int64_t q = 0; q < j; q++) s -= Aj[q] * Aj[q];
    if (!(s > 0.0)) return j + 1;
    double d = sqrt(s);
    Aj[j] = d;
    for (int64_t i = j + 1; i < n; i++) {
      double *Ai = A + i * n;
      double t = Ai[j];
      for (int64_t q = 0; q < j; q++) t -= Ai[q] * Aj[q];
      Ai[j] = t / d;
    }
    for (int64_t q = j + 1; q < n; q++) Aj[q] = 0.0;
  }
  return 0;
}

/* x := K^-1 b  via L L^T (gonum Cholesky.SolveVecTo) */
static void chol_solve_vec(const double *L, int64_t n, double *x) {
  for (int64_t i = 0; i < n; i++) {
    double t = x[i];
    const double *Li = L + i * n;
    for (int64_t q = 0; q < i; q++) t -= Li[q] * x[q];
    x[i] = t / Li[i];
  }
  for (int64_t i = n - 1; i >= 0; i--) {
    double t = x[i];
    for (int64_t q = i + 1; q < n; q++) t -= L[q * n + i] * x[q];
    x[i] = t / L[i * n + i];
  }
}

/* B (n x m, row-major) := K^-1 B  (gonum Cholesky.SolveTo) */
static void chol_solve_mat(const double *L, int64_t n, double *B, int64_t m) {
  for (int64_t i = 0; i < n; i++) {
    const double *Li = L + i * n;
    double *Bi = B + i * m;
    for (int64_t q = 0; q < i; q++) {
      double l = Li[q];
      if (l == 0.0) continue;
      const double *Bq = B + q * m;
      for (int64_t c = 0; c < m; c++) Bi[c] -= l * Bq[c];
    }
    double inv = 1.0 / Li[i];
    for (int64_t c = 0; c < m; c++) Bi[c] *= inv;
  }
  for (int64_t i = n - 1; i >= 0; i--) {
    double *Bi = B + i * m;
    for (int64_t q = i + 1; q < n; q++) {
      double l = L[q * n + i];
      if (l == 0.0) continue;
      const double *Bq = B + q * m;
      for (int64_t c = 0; c < m; c++) Bi[c] -= l * Bq[c];
    }
    double inv = 1.0 / L[i * n + i];
    for (int64_t c = 0; c < m; c++) Bi[c] *= inv;
  }
}

/* ---- absorb: gp/gp.go:89-239 ------------------------------------------------ */

/* gp/gp.go:61-72 addTodK, on dense symmetric storage */
static void add_to_dk(gogp_oracle *o, int64_t i, int64_t j, int64_t ipar0,
                      int64_t jpar0, int64_t narg, const double *grad) {
  int64_t n = o->n;
  for (int64_t a = 0; a < narg; a++) {
    double *M = o->dK[ipar0 + a];
    double v = M[i * n + j] + grad[jpar0 + a];
    M[i * n + j] = v; /* SetSym */
    M[j * n + i] = v;
  }
}

static int absorb(gogp_oracle *o, int with_grad, int64_t *pivot) {
  const gogp_desc *d = &o->desc;
  int64_t n = o->n;
  int D = d->ndim, ns = d->ntheta_simil, nn = ntheta_noise(d);
  free_dk(o);
  free(o->K);
  free(o->L);
  free(o->alpha);
  o->K = o->L = o->alpha = NULL;
  if (with_grad) { /* gp/gp.go:90-99 */
    o->ndk = ns + nn + (o->with_obs ? D * n : 0);
    o->dK = (double **)calloc((size_t)(o->ndk > 0 ? o->ndk : 1), sizeof(double *));
    o->has_dk = 1;
  }
  if (n == 0) return GOGP_OK; /* gp/gp.go:101-104 */
  if (with_grad)              /* gp/gp.go:158-163 */
    for (int64_t p = 0; p < o->ndk; p++)
      o->dK[p] = (double *)calloc((size_t)(n * n), sizeof(double));
  o->K = (double *)calloc((size_t)(n * n), sizeof(double));
  double *kgrad = (double *)malloc(sizeof(double) * (size_t)(ns + 2 * D + 1));
  double *ngrad = (double *)malloc(sizeof(double) * (size_t)(nn + D + 1));
  /* serial pair loop: gp/gp.go:220-225; the Parallel variant
   * (gp/gp.go:165-213) computes the same entries */
  for (int64_t i = 0; i < n; i++) {
    for (int64_t j = i; j < n; j++) {
      const double *xi = o->X + i * D, *xj = o->X + j * D;
      /* cov: gp/gp.go:109-156 */
      double k = gogp_oracle_simil(d, o->theta_s, xi, xj, with_grad ? kgrad : NULL);
      if (with_grad) {
        for (int a = 0; a < ns; a++) kgrad[a] *= o->theta_s[a]; /* :114-116 */
        add_to_dk(o, i, j, 0, 0, ns, kgrad);                    /* :117 */
        if (o->with_obs) {                                      /* :118-129 */
          add_to_dk(o, i, j, ns + nn + i * D, ns, D, kgrad);
          add_to_dk(o, i, j, ns + nn + j * D, ns + D, D, kgrad);
        }
      }
      if (j == i) { /* :133-154 */
        double nv = gogp_oracle_noise(d, o->theta_n, xj, with_grad ? ngrad : NULL);
        if (with_grad) {
          for (int a = 0; a < nn; a++) ngrad[a] *= o->theta_n[a];
          add_to_dk(o, i, j, ns, 0, nn, ngrad);
          if (o->with_obs) add_to_dk(o, i, j, ns + nn + j * D, nn, D, ngrad);
        }
        k += nv;
      }
      o->K[i * n + j] = k; /* SetSym, :155 */
      o->K[j * n + i] = k;
    }
  }
  free(kgrad);
  free(ngrad);
  /* gp/gp.go:228-230 */
  o->L = (double *)malloc(sizeof(double) * (size_t)(n * n));
  memcpy(o->L, o->K, sizeof(double) * (size_t)(n * n));
  int64_t bad = chol_lower(o->L, n);
  if (bad) {
    if (pivot) *pivot = bad - 1;
    free(o->L);
    o->L = NULL;
    return GOGP_ENOTPD;
  }
  /* gp/gp.go:232-236 */
  o->alpha = (double *)malloc(sizeof(double) * (size_t)n);
  memcpy(o->alpha, o->Y, sizeof(double) * (size_t)n);
  chol_solve_vec(o->L, n, o->alpha);
  return GOGP_OK;
}

static int set_data(gogp_oracle *o, const double *X, const double *y, int64_t n) {
  int D = o->desc.ndim;
  free(o->X);
  free(o->Y);
  o->X = (double *)malloc(sizeof(double) * (size_t)(n * D + 1));
  o->Y = (double *)malloc(sizeof(double) * (size_t)(n + 1));
  if (n > 0) {
    memcpy(o->X, X, sizeof(double) * (size_t)(n * D));
    memcpy(o->Y, y, sizeof(double) * (size_t)n);
  }
  o->n = n;
  return GOGP_OK;
}

/* gp.GP.Absorb, gp/gp.go:80-87 */
int gogp_oracle_absorb(gogp_oracle *o, const double *theta_s, const double *theta_n,
                       const double *X, const double *y, int64_t n, int64_t *pivot) {
  if (theta_s) memcpy(o->theta_s, theta_s, sizeof(double) * (size_t)o->desc.ntheta_simil);
  if (theta_n) memcpy(o->theta_n, theta_n, sizeof(double) * (size_t)ntheta_noise(&o->desc));
  set_data(o, X, y, n);
  o->with_obs = 0;
  return absorb(o, 0, pivot);
}

/* gp.GP.LML, gp/gp.go:244-253 */
double gogp_oracle_lml(const gogp_oracle *o) {
  double lml = 0.0;
  int64_t n = o->n;
  if (n == 0 || !o->L) return lml;
  lml -= 0.5 * (double)n * log(2 * M_PI);
  double logdet = 0.0; /* gonum Cholesky.LogDet = 2 sum log U_ii */
  for (int64_t i = 0; i < n; i++) logdet += 2.0 * log(o->L[i * n + i]);
  lml -= 0.5 * logdet;
  double dot = 0.0;
  for (int64_t i = 0; i < n; i++) dot += o->Y[i] * o->alpha[i];
  lml -= 0.5 * dot;
  return lml;
}

/* gp.GP.Observe, gp/gp.go:374-413.  x is mutated exactly like the reference:
 * x[:P] is exp()ed in place and log()ed back (:378-381,:408-410).  When
 * len == P the data must have been assigned before (Xset/yset, nset). */
int gogp_oracle_observe(gogp_oracle *o, double *x, int64_t len, const double *Xset,
                        const double *yset, int64_t nset, double *lml,
                        int64_t *pivot) {
  const gogp_desc *d = &o->desc;
  int D = d->ndim, ns = d->ntheta_simil, nn = ntheta_noise(d);
  int P = ns + nn;
  if (len < P) return GOGP_EARG;
  for (int i = 0; i < P; i++) x[i] = exp(x[i]); /* :378-381 */
  memcpy(o->theta_s, x, sizeof(double) * (size_t)ns);      /* :384 */
  memcpy(o->theta_n, x + ns, sizeof(double) * (size_t)nn); /* :385 */
  int64_t rest = len - P;
  o->with_obs = rest > 0; /* :386 */
  int rc = GOGP_OK;
  if (o->with_obs) {
    int64_t n = rest / (D + 1); /* :391 */
    if (n * (D + 1) != rest) rc = GOGP_EARG; /* panic("len(x)") :398-400 */
    else set_data(o, x + P, x + P + n * D, n);
  } else {
    set_data(o, Xset, yset, nset);
  }
  if (rc == GOGP_OK) rc = absorb(o, 1, pivot); /* :402 */
  for (int i = 0; i < P; i++) x[i] = log(x[i]); /* :408-410 */
  if (rc != GOGP_OK) return rc;
  if (lml) *lml = gogp_oracle_lml(o);
  return GOGP_OK;
}

/* gp.GP.Gradient, gp/gp.go:418-499: per parameter
 *   r0 = (alpha alpha^T) dK ; r1 = K^-1 dK ; grad = 1/2 tr(r0 - r1) */
int gogp_oracle_gradient(gogp_oracle *o, double *grad, int64_t len) {
  const gogp_desc *d = &o->desc;
  int64_t n = o->n;
  int D = d->ndim, P = d->ntheta_simil + ntheta_noise(d);
  int64_t want = o->with_obs ? P + n * (D + 1) : P; /* :420-425 */
  if (len != want) return GOGP_EARG;
  for (int64_t i = 0; i < len; i++) grad[i] = 0.0;
  if (n == 0) return GOGP_OK; /* :427-430 */
  if (!o->has_dk || !o->L) return GOGP_ESTATE;
  double *a = (double *)malloc(sizeof(double) * (size_t)(n * n));
  double *r0 = (double *)malloc(sizeof(double) * (size_t)(n * n));
  double *r1 = (double *)malloc(sizeof(double) * (size_t)(n * n));
  for (int64_t i = 0; i < n; i++) /* :434-435 */
    for (int64_t j = 0; j < n; j++) a[i * n + j] = o->alpha[i] * o->alpha[j];
  for (int64_t p = 0; p < o->ndk; p++) { /* :476-485 */
    const double *dKp = o->dK[p];
    for (int64_t i = 0; i < n; i++)
      for (int64_t j = 0; j < n; j++) {
        double s = 0.0;
        for (int64_t q = 0; q < n; q++) s += a[i * n + q] * dKp[q * n + j];
        r0[i * n + j] = s;
      }
    memcpy(r1, dKp, sizeof(double) * (size_t)(n * n));
    chol_solve_mat(o->L, n, r1, n);
    double tr = 0.0;
    for (int64_t i = 0; i < n; i++) tr += r0[i * n + i] - r1[i * n + i];
    grad[p] = 0.5 * tr;
  }
  if (o->with_obs) /* :488-493 */
    for (int64_t i = 0; i < n; i++) grad[o->ndk + i] = -o->alpha[i];
  free(a);
  free(r0);
  free(r1);
  free_dk(o); /* :496 */
  return GOGP_OK;
}

/* gp.GP.Produce, gp/gp.go:258-360 */
int gogp_oracle_produce(gogp_oracle *o, const double *Z, int64_t m, double *mu,
                        double *sigma) {
  const gogp_desc *d = &o->desc;
  int64_t n = o->n;
  int D = d->ndim;
  double *variance = (double *)malloc(sizeof(double) * (size_t)(m + 1));
  double *covdiag = (double *)calloc((size_t)(m + 1), sizeof(double));
  for (int64_t i = 0; i < m; i++) /* :269-278 prior variance, Simil only */
    variance[i] = gogp_oracle_simil(d, o->theta_s, Z + i * D, Z + i * D, NULL);
  for (int64_t i = 0; i < m; i++) mu[i] = 0.0;
  if (n > 0) { /* :282-342 */
    if (!o->L) {
      free(variance);
      free(covdiag);
      return GOGP_ESTATE;
    }
    double *Kstar = (double *)malloc(sizeof(double) * (size_t)(n * m));
    for (int64_t i = 0; i < n; i++) /* :322-332 */
      for (int64_t j = 0; j < m; j++)
        Kstar[i * m + j] = gogp_oracle_simil(d, o->theta_s, o->X + i * D, Z + j * D, NULL);
    for (int64_t j = 0; j < m; j++) { /* :335 mean = Kstar^T alpha */
      double s = 0.0;
      for (int64_t i = 0; i < n; i++) s += Kstar[i * m + j] * o->alpha[i];
      mu[j] = s;
    }
    double *v = (double *)malloc(sizeof(double) * (size_t)(n * m));
    memcpy(v, Kstar, sizeof(double) * (size_t)(n * m));
    chol_solve_mat(o->L, n, v, m); /* :337-340 */
    for (int64_t j = 0; j < m; j++) { /* :341-342, only the diagonal is read :356 */
      double s = 0.0;
      for (int64_t i = 0; i < n; i++) s += Kstar[i * m + j] * v[i * m + j];
      covdiag[j] = s;
    }
    free(Kstar);
    free(v);
  }
  for (int64_t i = 0; i < m; i++) /* :354-357, unclamped */
    sigma[i] = sqrt(variance[i] - covdiag[i]);
  free(variance);
  free(covdiag);
  return GOGP_OK;
}

/* ---- accessors --------------------------------------------------------------*/
int64_t gogp_oracle_n(const gogp_oracle *o) { return o->n; }
const double *gogp_oracle_alpha(const gogp_oracle *o) { return o->alpha; }
const double *gogp_oracle_factor(const gogp_oracle *o) { return o->L; }
const double *gogp_oracle_gram(const gogp_oracle *o) { return o->K; }
const double *gogp_oracle_dk(const gogp_oracle *o, int64_t p) {
  return (o->dK && p < o->ndk) ? o->dK[p] : NULL;
}
int64_t gogp_oracle_ndk(const gogp_oracle *o) { return o->ndk; }
const double *gogp_oracle_theta_simil(const gogp_oracle *o) { return o->theta_s; }
const double *gogp_oracle_theta_noise(const gogp_oracle *o) { return o->theta_n; }

/* ---- multi-threaded O(N^2) pieces for the "fast" CPU path --------------------
 * Same mathematics as absorb()/gradient above in the W-matrix form
 *     grad_p = 1/2 sum_ij (alpha alpha^T - K^-1)_ij * theta_p dK_ij/dtheta_p ,
 * used by oracle.FastOracle (with LAPACK potrf/potri for the O(N^3) parts) at
 * sizes where the faithful per-parameter dense products (gp/gp.go:476-485) are
 * out of reach, and as bench.py's cpu_baseline ("port").  OpenMP over rows. */

/* K (n x n, full symmetric) = Simil(x_i,x_j) + [i==j] noise_var : gp/gp.go:109-156 */
void gogp_oracle_gram_omp(const gogp_desc *d, const double *theta_s, double noise_var,
                          const double *X, int64_t n, double *K) {
  int D = d->ndim;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < n; i++) {
    for (int64_t j = 0; j <= i; j++) {
      double k = gogp_oracle_simil(d, theta_s, X + i * D, X + j * D, NULL);
      if (i == j) k += noise_var;
      K[i * n + j] = k;
      K[j * n + i] = k;
    }
  }
}

/* Kstar (n x m) = Simil(x_i, z_j) : gp/gp.go:322-332 */
void gogp_oracle_cross_omp(const gogp_desc *d, const double *theta_s, const double *X,
                           int64_t n, const double *Z, int64_t m, double *Ks) {
  int D = d->ndim;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    for (int64_t j = 0; j < m; j++)
      Ks[i * m + j] = gogp_oracle_simil(d, theta_s, X + i * D, Z + j * D, NULL);
}

/* out[p] (p < ntheta_simil) = 1/2 sum_ij W_ij theta_p dk_ij/dtheta_p,
 * out[ntheta_simil] = trace(W);  W = alpha alpha^T - Kinv, Kinv read from its
 * lower triangle (row-major n x n). */
void gogp_oracle_grad_reduce_omp(const gogp_desc *d, const double *theta_s, const double *X,
                                 const double *alpha, const double *Kinv, int64_t n,
                                 double *out) {
  int D = d->ndim, ns = d->ntheta_simil;
  for (int p = 0; p <= ns; p++) out[p] = 0.0;
#pragma omp parallel
  {
    double acc[GOGP_MAX_NDIM + 16];
    double g[3 * GOGP_MAX_NDIM + 16];
    for (int p = 0; p <= ns; p++) acc[p] = 0.0;
#pragma omp for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; i++) {
      for (int64_t j = 0; j <= i; j++) {
        double w = alpha[i] * alpha[j] - Kinv[i * n + j];
        double wgt = (j < i) ? 2.0 * w : w;
        gogp_oracle_simil(d, theta_s, X + i * D, X + j * D, g);
        for (int p = 0; p < ns; p++) acc[p] += wgt * g[p] * theta_s[p];
        if (i == j) acc[ns] += w;
      }
    }
#pragma omp critical
    for (int p = 0; p <= ns; p++) out[p] += (p < ns ? 0.5 : 1.0) * acc[p];
  }
}
