// Drives the C++ host mirror (gogp_amd/host/gogp.hpp) through the reference's
// "noise" TestProduce case (gp/gp_test.go:107-120) and the "uninoise"
// TestElementalModel case (gp/gp_test.go:220-229, hyperparameters-only form).
// Exit code 0 = all within the reference's 1e-6; 3 = no HIP device (GOGP_EHIP).
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../gogp_amd/host/gogp.hpp"

static gogp_desc normal_desc(int noise_kind, double std) {
  gogp_desc d;
  std::memset(&d, 0, sizeof d);
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

int main() {
  try {
    gogp::GP gp(normal_desc(GOGP_NOISE_CONSTANT, 0.1));
    gp.ThetaSimil = {1.0};
    if (gp.Absorb({{0.0}, {1.0}}, {1.0, -1.0}) != GOGP_OK) return 1;
    std::vector<double> mu, sigma;
    if (!gp.Produce({{-2.0}, {3.0}}, mu, sigma)) return 1;
    const double wmu[2] = {0.307895, -0.307895}, wsig[2] = {0.987037, 0.987037};
    for (int i = 0; i < 2; ++i)
      if (std::fabs(mu[i] - wmu[i]) > 1e-6 || std::fabs(sigma[i] - wsig[i]) > 1e-6) {
        std::printf("produce mismatch %d: %g %g\n", i, mu[i], sigma[i]);
        return 1;
      }
    if (gp.Alpha.size() != 2) return 1;  // refreshed by Absorb (gp/gp.go:35-36)
    gogp::GP g2(normal_desc(GOGP_NOISE_UNIFORM, 0.0));
    g2.SetData({{-1.0}, {-1.0}}, {1.0, 0.0});  // x = [1, 1 | -1, -1 | 1, 0]
    const double ll = g2.Observe({1.0, 1.0});
    const std::vector<double> a1 = g2.Alpha;
    // hyperparameter steps on resident data: no upload, Alpha follows
    const double llb = g2.Observe({0.7, 1.2});
    if (llb == ll || g2.Alpha == a1) return 1;
    if (g2.Observe({1.0, 1.0}) != ll || g2.Alpha != a1) return 1;
    // in-place edit of Y + Touch(): the change reaches the device
    g2.Y[1] = 0.5;
    g2.Touch();
    if (g2.Observe({1.0, 1.0}) == ll) return 1;
    g2.Y[1] = 0.0;
    g2.Touch();
    if (g2.Observe({1.0, 1.0}) != ll) return 1;
    const std::vector<double> Lf = g2.Factor();
    if (Lf.size() != 4 || Lf[1] != 0.0 || !(Lf[0] > 0.0)) return 1;
    if (std::fabs(ll - (-4.018110)) >= 1e-6) {
      std::printf("lml mismatch: %.9f\n", ll);
      return 1;
    }
    std::vector<double> g = g2.Gradient();
    if (g.size() != 2) return 1;
    // two candidates in one launch sequence: the current point and another one; the first must
    // reproduce Observe + Gradient bit for bit and the GP's own state must survive
    const double ll07 = g2.Observe({0.7, 1.2});
    if (g2.Observe({1.0, 1.0}) != ll) return 1;
    const gogp::GP::Candidates cd = g2.ObserveGradientCandidates({{1.0, 1.0}, {0.7, 1.2}});
    if (cd.lml.size() != 2 || cd.lml[0] != ll || cd.lml[1] != ll07 || cd.status[0] != 0 || cd.status[1] != 0) return 1;
    if (cd.grad[0] != g || g2.Gradient() != g) return 1;
    bool threw = false;
    try {
      g2.Observe({1.0, 1.0, 0.5});  // leftover 1 is not a multiple of NDim+1: gp/gp.go:398-400
    } catch (const gogp::Error &) {
      threw = true;
    }
    if (!threw) return 1;
    std::printf("cpp host ok: lml=%.6f grad=[%.6f %.6f]\n", ll, g[0], g[1]);
    return 0;
  } catch (const gogp::Error &e) {
    std::printf("gogp::Error %d: %s\n", e.code, e.what());
    return e.code == GOGP_EHIP ? 3 : 2;
  }
}
