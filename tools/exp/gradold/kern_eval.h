// kern_eval.h -- device evaluation of the similarity kernel and of its
// derivatives w.r.t. the LOG of every hyperparameter.
//
// Formulas: reference kernel/kernel.go:23-26 (Normal), :44-47 (Periodic),
// :70-73 (Matern32), :89-92 (Matern52, whose 5/3 is Go integer division = 1),
// with r^2 = sum_d ((xa_d - xb_d)/l_d)^2 for NDim > 1.  The derivatives replace
// infergo's per-pair AD tape (gp/gp.go:113-117): the reference multiplies each
// tape gradient by theta_p (gp/gp.go:114-116), i.e. differentiates w.r.t.
// log theta_p, which is what these closed forms return.
#pragma once
#include "common.h"

namespace gogp {

#define GOGP_SQRT3 1.7320508075688772  // kernel/kernel.go:51
#define GOGP_SQRT5 2.2360679774997900  // kernel/kernel.go:52
#define GOGP_PI 3.14159265358979323846

// f(r2) and df/d(r2) of the radial kinds
__device__ __forceinline__ void radial_eval(int kind, double r2, double &f, double &dfdr2) {
  if (kind == GOGP_K_NORMAL) {
    f = exp(-0.5 * r2);
    dfdr2 = -0.5 * f;
  } else {
    const double r = sqrt(r2);
    if (kind == GOGP_K_MATERN32) {
      const double e = exp(-GOGP_SQRT3 * r);
      f = (1.0 + GOGP_SQRT3 * r) * e;
      dfdr2 = -1.5 * e;
    } else if (kind == GOGP_K_MATERN52) {
      const double e = exp(-GOGP_SQRT5 * r);
      f = (1.0 + GOGP_SQRT5 * r + r2) * e;
      dfdr2 = -0.5 * (3.0 + GOGP_SQRT5 * r) * e;
    } else {  // GOGP_K_MATERN52_TEXTBOOK
      const double e = exp(-GOGP_SQRT5 * r);
      f = (1.0 + GOGP_SQRT5 * r + (5.0 / 3.0) * r2) * e;
      dfdr2 = -(5.0 / 6.0) * (1.0 + GOGP_SQRT5 * r) * e;
    }
  }
}

// value only.  xa(d), xb(d): accessors of the two inputs' coordinates.
template <class FA, class FB>
__device__ __forceinline__ double simil_value(const DevParams &P, FA xa, FB xb) {
  const int D = P.ndim;
  double k = 0.0;
  for (int t = 0; t < P.nterms; ++t) {
    const int kind = P.kind[t];
    double s = 0.0;
    if (kind == GOGP_K_PERIODIC) {
      const double w = P.w[t];
      for (int d = 0; d < D; ++d) {
        const double dd = sin(w * fabs(xa(d) - xb(d))) * P.inv_len[t][d];
        s += dd * dd;
      }
      k += P.c[t] * exp(-2.0 * s);
    } else {
      for (int d = 0; d < D; ++d) {
        const double u = (xa(d) - xb(d)) * P.inv_len[t][d];
        s += u * u;
      }
      double f, dfdr2;
      radial_eval(kind, s, f, dfdr2);
      k += P.c[t] * f;
    }
  }
  return k;
}

// Accumulate  wgt * (theta_p dk/dtheta_p)  into the slot accumulators:
//   acc[3t+0] scale, acc[3t+1] length (non-ARD), acc[3t+2] period,
//   ard[d] per-dimension length of the (single) ARD term.
template <int ARD_D, class FA, class FB>
__device__ __forceinline__ void simil_grad_accum(const DevParams &P, FA xa, FB xb, double wgt,
                                                 double *acc, double *ard) {
  const int D = P.ndim;
#pragma unroll
  for (int t = 0; t < GOGP_MAX_TERMS; ++t) {
    if (t >= P.nterms) break;
    const int kind = P.kind[t];
    const double c = P.c[t];
    if (kind == GOGP_K_PERIODIC) {
      const double w = P.w[t];
      double s = 0.0, gp = 0.0;
      for (int d = 0; d < D; ++d) {
        const double phi = w * fabs(xa(d) - xb(d));
        double sn, cs;
        sincos(phi, &sn, &cs);
        const double il = P.inv_len[t][d];
        const double dd = sn * il;
        s += dd * dd;
        gp += dd * cs * phi * il;
      }
      const double f = exp(-2.0 * s);
      const double cf = wgt * c * f;
      acc[3 * t + 0] += cf;
      acc[3 * t + 2] += cf * 4.0 * gp;
      if (P.ard[t]) {
        if (ARD_D > 0) {
#pragma unroll
          for (int d = 0; d < ARD_D; ++d)
            if (d < D) {
              const double dd = sin(w * fabs(xa(d) - xb(d))) * P.inv_len[t][d];
              ard[d] += cf * 4.0 * dd * dd;
            }
        }
      } else {
        acc[3 * t + 1] += cf * 4.0 * s;
      }
    } else {
      double s = 0.0;
      for (int d = 0; d < D; ++d) {
        const double u = (xa(d) - xb(d)) * P.inv_len[t][d];
        s += u * u;
      }
      double f, dfdr2;
      radial_eval(kind, s, f, dfdr2);
      acc[3 * t + 0] += wgt * c * f;
      const double g = wgt * c * dfdr2 * (-2.0);
      if (P.ard[t]) {
        if (ARD_D > 0) {
#pragma unroll
          for (int d = 0; d < ARD_D; ++d)
            if (d < D) {
              const double u = (xa(d) - xb(d)) * P.inv_len[t][d];
              ard[d] += g * u * u;
            }
        }
      } else {
        acc[3 * t + 1] += g * s;
      }
    }
  }
}

// Accumulate  W * dk(xa, xb)/dxa_d  into acc[d]  (d < ndim <= DMAX): the
// derivative the AD tape yields for the first input (gp/gp.go:118-123).  For the
// stationary kernels here dk/dxb = -dk/dxa.
template <int DMAX, class FA, class FB>
__device__ __forceinline__ void simil_xgrad_accum(const DevParams &P, FA xa, FB xb, double W,
                                                  double *acc) {
  const int D = P.ndim;
  for (int t = 0; t < P.nterms; ++t) {
    const int kind = P.kind[t];
    const double c = P.c[t];
    if (kind == GOGP_K_PERIODIC) {
      const double w = P.w[t];
      double s = 0.0;
      for (int d = 0; d < D; ++d) {
        const double dd = sin(w * fabs(xa(d) - xb(d))) * P.inv_len[t][d];
        s += dd * dd;
      }
      const double cf = W * c * exp(-2.0 * s) * (-4.0) * w;
#pragma unroll
      for (int d = 0; d < DMAX; ++d)
        if (d < D) {
          const double dx = xa(d) - xb(d);
          const double phi = w * fabs(dx);
          double sn, cs;
          sincos(phi, &sn, &cs);
          const double il = P.inv_len[t][d];
          const double sg = dx > 0.0 ? 1.0 : (dx < 0.0 ? -1.0 : 0.0);
          acc[d] += cf * (sn * il) * cs * il * sg;
        }
    } else {
      double s = 0.0;
      for (int d = 0; d < D; ++d) {
        const double u = (xa(d) - xb(d)) * P.inv_len[t][d];
        s += u * u;
      }
      double f, dfdr2;
      radial_eval(kind, s, f, dfdr2);
      const double g = W * c * dfdr2 * 2.0;
#pragma unroll
      for (int d = 0; d < DMAX; ++d)
        if (d < D) {
          const double il = P.inv_len[t][d];
          acc[d] += g * (xa(d) - xb(d)) * il * il;
        }
    }
  }
}

}  // namespace gogp
