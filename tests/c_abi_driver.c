/* Plain C11 driver of the C ABI (include/gogp_hip.h) in exactly the order the cgo shim
 * (go/gogp/gp.go) issues the calls:
 *     gogp_create -> gogp_set_data ONCE -> (gogp_observe -> gogp_gradient) x k with changing
 *     hyperparameters -> gogp_get_alpha -> gogp_produce -> gogp_absorb -> gogp_lml ->
 *     gogp_get_factor_diag -> gogp_destroy
 * Checks the reference's known answers (gp/gp_test.go:107-120 "noise", :220-229 "uninoise")
 * to its 1e-6, that repeated Observe calls need no new gogp_set_data, the third noise kind
 * (tutorial/anynoise/kernel/kernel.go:26-35: constant variance WITH a parameter) and the
 * error codes.  Exit 0 = ok, 3 = no HIP device (GOGP_EHIP: there is no CPU fallback). */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "../include/gogp_hip.h"

static gogp_desc normal_desc(int noise_kind, double std) {
  gogp_desc d;
  memset(&d, 0, sizeof d);
  d.ndim = 1;
  d.nterms = 1;
  d.ntheta_simil = 1;
  d.noise_kind = noise_kind;
  d.noise_std = std;
  d.noise_scale = 1.0;
  d.terms[0].kind = GOGP_K_NORMAL;
  d.terms[0].scale_idx = -1;
  d.terms[0].len_idx = 0;
  d.terms[0].period_idx = -1;
  d.terms[0].period_mult = 1.0;
  return d;
}

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc_ = (call);                                                        \
    if (rc_ != GOGP_OK) {                                                    \
      printf("%s -> %d: %s\n", #call, rc_, gogp_last_error(h));              \
      return rc_ == GOGP_EHIP ? 3 : 2;                                       \
    }                                                                        \
  } while (0)

int main(void) {
  gogp_handle *h = NULL;
  gogp_desc d = normal_desc(GOGP_NOISE_UNIFORM, 0.0);
  if (gogp_desc_check(&d) != GOGP_OK || gogp_desc_ntheta_noise(&d) != 1) return 2;
  int rc = gogp_create(&d, -1, &h);
  if (rc != GOGP_OK) {
    printf("gogp_create -> %d: %s\n", rc, gogp_last_error(NULL));
    return rc == GOGP_EHIP ? 3 : 2;
  }
  /* "uninoise" (gp/gp_test.go:220-229): x = [1, 1 | -1, -1 | 1, 0] in the hyperparameters-only form */
  const double X[2] = {-1.0, -1.0}, y[2] = {1.0, 0.0};
  CHECK(gogp_set_data(h, X, y, 2)); /* once */
  double lml = 0, grad[2], grad2[2], alpha[2];
  const double x1[2] = {1.0, 1.0}, x2[2] = {0.7, 1.2};
  CHECK(gogp_observe(h, x1, 2, &lml));
  if (fabs(lml - (-4.018110)) >= 1e-6) return printf("lml %.9f\n", lml), 1;
  CHECK(gogp_gradient(h, grad, 2));
  double lml2 = 0, lml3 = 0;
  CHECK(gogp_observe(h, x2, 2, &lml2)); /* other hyperparameters, same resident data */
  CHECK(gogp_gradient(h, grad2, 2));
  CHECK(gogp_observe(h, x1, 2, &lml3)); /* and back: identical result, no set_data in between */
  if (lml3 != lml || lml2 == lml) return printf("repeat %.17g %.17g %.17g\n", lml, lml2, lml3), 1;
  /* the same two points as candidates of ONE launch sequence: bit-equal, handle state untouched */
  {
    const double xs[4] = {1.0, 1.0, 0.7, 1.2};
    double lmls[2], grads[4], gnow[2];
    int st[2] = {-1, -1};
    CHECK(gogp_observe_gradient_candidates(h, 2, xs, 2, lmls, grads, st));
    if (st[0] != GOGP_OK || st[1] != GOGP_OK || lmls[0] != lml || lmls[1] != lml2)
      return printf("candidates %.17g %.17g vs %.17g %.17g\n", lmls[0], lmls[1], lml, lml2), 1;
    if (grads[0] != grad[0] || grads[1] != grad[1] || grads[2] != grad2[0] || grads[3] != grad2[1])
      return printf("candidate gradients differ\n"), 1;
    CHECK(gogp_gradient(h, gnow, 2)); /* still the gradient of the last gogp_observe (x1) */
    if (gnow[0] != grad[0] || gnow[1] != grad[1]) return printf("handle state changed\n"), 1;
    if (gogp_observe_gradient_candidates(h, 2, xs, 3, lmls, grads, st) != GOGP_EARG) return 1;
    if (gogp_observe_gradient_candidates(h, GOGP_MAX_CANDIDATES + 1, xs, 2, lmls, grads, st) != GOGP_EARG) return 1;
  }
  CHECK(gogp_get_alpha(h, alpha));
  if (gogp_n(h) != 2) return 1;
  if (gogp_gradient(h, grad, 3) != GOGP_EARG) return printf("gradient length not checked\n"), 1;
  if (gogp_observe(h, x1, 3, &lml) != GOGP_EARG) return printf("len(x) not checked\n"), 1; /* gp/gp.go:398-400 */
  gogp_destroy(h);

  /* "noise" TestProduce case (gp/gp_test.go:107-120): Absorb + Produce, ConstantNoise(0.1) */
  d = normal_desc(GOGP_NOISE_CONSTANT, 0.1);
  CHECK(gogp_create(&d, -1, &h));
  const double Xa[2] = {0.0, 1.0}, ya[2] = {1.0, -1.0}, Z[2] = {-2.0, 3.0};
  const double ts[1] = {1.0}, tn[1] = {0.0};
  CHECK(gogp_set_data(h, Xa, ya, 2));
  CHECK(gogp_absorb(h, ts, tn));
  double mu[2], sigma[2], diag[2];
  CHECK(gogp_produce(h, Z, 2, mu, sigma));
  const double wmu[2] = {0.307895, -0.307895}, wsig[2] = {0.987037, 0.987037};
  for (int i = 0; i < 2; ++i)
    if (fabs(mu[i] - wmu[i]) > 1e-6 || fabs(sigma[i] - wsig[i]) > 1e-6)
      return printf("produce %d: %g %g\n", i, mu[i], sigma[i]), 1;
  CHECK(gogp_lml(h, &lml));
  CHECK(gogp_get_factor_diag(h, diag));
  CHECK(gogp_get_alpha(h, alpha));
  /* LML = -n/2 log 2pi - sum log L_ii - y.alpha/2 (gp/gp.go:244-253) from the exported state */
  const double lml_host = -log(2 * 3.14159265358979323846) - log(diag[0]) - log(diag[1]) -
                          0.5 * (ya[0] * alpha[0] + ya[1] * alpha[1]);
  if (fabs(lml_host - lml) > 1e-12 * fabs(lml)) return printf("lml %.12f vs %.12f\n", lml, lml_host), 1;
  if (gogp_gradient(h, grad, 1) != GOGP_ESTATE) return printf("Gradient after Absorb must fail\n"), 1;
  gogp_destroy(h);

  /* constant noise that owns a parameter (tutorial/anynoise): NTheta = 1, gradient component 0 */
  d = normal_desc(GOGP_NOISE_CONSTANT_PARAM, 0.1);
  if (gogp_desc_ntheta_noise(&d) != 1) return 1;
  CHECK(gogp_create(&d, -1, &h));
  CHECK(gogp_set_data(h, Xa, ya, 2));
  const double xa[2] = {0.0, 5.0}, xb[2] = {0.0, -3.0};
  double la = 0, lb = 0;
  CHECK(gogp_observe(h, xa, 2, &la));
  CHECK(gogp_gradient(h, grad, 2));
  CHECK(gogp_observe(h, xb, 2, &lb));
  if (la != lb || grad[1] != 0.0) return printf("constant-param noise: %g %g %g\n", la, lb, grad[1]), 1;
  /* not positive definite: duplicate inputs, zero noise (gp/gp.go:228-230) */
  gogp_destroy(h);
  d = normal_desc(GOGP_NOISE_CONSTANT, 0.0);
  CHECK(gogp_create(&d, -1, &h));
  const double Xd[3] = {0.0, 0.0, 1.0}, yd[3] = {1.0, 1.0, 0.0};
  CHECK(gogp_set_data(h, Xd, yd, 3));
  if (gogp_absorb(h, ts, tn) != GOGP_ENOTPD || gogp_notpd_index(h) != 1)
    return printf("not-PD not reported: pivot %lld\n", (long long)gogp_notpd_index(h)), 1;
  gogp_destroy(h);
  printf("c abi ok: lml=%.6f grad=[%.6f %.6f]\n", lml3, grad2[0], grad2[1]);
  return 0;
}
