// gogp.hpp -- C++ host-side mirror of the reference's gp.GP (gp/gp.go:20-38) over
// the C ABI of include/gogp_hip.h.  Header-only; link with -lgogp_hip.
//
// The reference is Go (compiled code) and no Go toolchain exists in the build
// image, so this header is the compiled-language host layer; the Go shim a
// maintainer would add is in go/gogp/gp.go.  Method names,
// argument meaning and error behaviour follow the reference: Absorb returns an
// error code (gp/gp.go:228-230), Observe throws where the reference panics
// (gp/gp.go:398-405), Produce returns false on error (gp/gp.go:338-340).
//
// X / Y are uploaded only when they changed: assign them through SetData() (or Absorb);
// after writing into the public X / Y members directly call Touch().  Alpha is refreshed
// after every Absorb / Observe, L lazily by Factor() (the reference documents L, Alpha, X,
// ThetaSimil, ThetaNoise as the state Produce depends on: gp/gp.go:35-36,255-257).
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gogp_hip.h"

namespace gogp {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

class GP {
 public:
  // Configuration (gp/gp.go:22-23)
  int NDim;
  gogp_desc desc;  // Simil + Noise in closed form (see gogp_desc)
  // Data (gp/gp.go:26-28)
  std::vector<double> ThetaSimil, ThetaNoise;
  std::vector<std::vector<double>> X;
  std::vector<double> Y;
  bool Parallel = false;  // accepted for compatibility
  // Cached computations (gp/gp.go:34-36)
  std::vector<double> Alpha;  // K^-1 y, refreshed by Absorb / Observe

  explicit GP(const gogp_desc &d, int device = -1) : NDim(d.ndim), desc(d) {
    if (gogp_create(&desc, device, &h_) != GOGP_OK)
      throw Error(GOGP_EHIP, gogp_last_error(nullptr));
    ThetaSimil.assign(desc.ntheta_simil, 0.0);  // gp/gp.go:50-56
    ThetaNoise.assign(gogp_desc_ntheta_noise(&desc), 0.0);
  }
  ~GP() { gogp_destroy(h_); }
  GP(const GP &) = delete;
  GP &operator=(const GP &) = delete;

  // gp/gp.go:80-87
  int Absorb(const std::vector<std::vector<double>> &x, const std::vector<double> &y) {
    SetData(x, y);
    int rc = push();
    if (rc != GOGP_OK) return rc;
    double zero = 0.0;
    rc = gogp_absorb(h_, ThetaSimil.data(), ThetaNoise.empty() ? &zero : ThetaNoise.data());
    if (rc != GOGP_OK && rc != GOGP_ECOND) return rc;  // gp/gp.go:228-230
    const int ra = fetch_alpha();
    return ra != GOGP_OK ? ra : rc;  // GOGP_ECOND: gonum's Condition error, gp/gp.go:233-236
  }

  // Assign the observations (gp.GP.X / gp.GP.Y); uploaded on the next Absorb / Observe.
  void SetData(const std::vector<std::vector<double>> &x, const std::vector<double> &y) {
    X = x;
    Y = y;
    dirty_ = true;
  }
  // Call after modifying the public X / Y members in place.
  void Touch() { dirty_ = true; }

  // gp/gp.go:244-253
  double LML() {
    double v = 0;
    check(gogp_lml(h_, &v));
    return v;
  }

  // gp/gp.go:258-360
  bool Produce(const std::vector<std::vector<double>> &x, std::vector<double> &mu,
               std::vector<double> &sigma) {
    std::vector<double> flat = pack(x);
    mu.assign(x.size(), 0.0);
    sigma.assign(x.size(), 0.0);
    return gogp_produce(h_, flat.data(), (int64_t)x.size(), mu.data(), sigma.data()) == GOGP_OK;
  }

  // gp/gp.go:374-413
  double Observe(const std::vector<double> &x) {
    const size_t P = ThetaSimil.size() + ThetaNoise.size();
    if (x.size() < P) throw Error(GOGP_EARG, "len(x)");
    double lml = 0;
    if (x.size() == P) {
      check(push());  // uploads only if X / Y changed since the last call
      check(gogp_observe(h_, x.data(), (int64_t)x.size(), &lml));
    } else {
      if ((x.size() - P) % (NDim + 1)) throw Error(GOGP_EARG, "len(x)");  // gp/gp.go:398-400
      const int rc = gogp_observe_full(h_, x.data(), (int64_t)x.size(), &lml);
      // gp/gp.go:391-396: X, Y are re-sliced from x (the device holds them now)
      const size_t n = (x.size() - P) / (size_t)(NDim + 1);
      X.assign(n, std::vector<double>((size_t)NDim));
      for (size_t i = 0; i < n; ++i)
        for (int d = 0; d < NDim; ++d) X[i][d] = x[P + i * NDim + d];
      Y.assign(x.begin() + (long)(P + n * NDim), x.end());
      dirty_ = rc != GOGP_OK;
      check(rc);
    }
    check(fetch_alpha());
    for (size_t i = 0; i < ThetaSimil.size(); ++i) ThetaSimil[i] = std::exp(x[i]);
    for (size_t i = 0; i < ThetaNoise.size(); ++i) ThetaNoise[i] = std::exp(x[ThetaSimil.size() + i]);
    last_len_ = x.size();
    return lml;
  }

  // gp/gp.go:418-499
  std::vector<double> Gradient() {
    std::vector<double> g(last_len_, 0.0);
    check(gogp_gradient(h_, g.data(), (int64_t)g.size()));
    return g;
  }

  // k candidate log-theta vectors (hyperparameters-only form) on this GP's data in ONE launch
  // sequence (gogp_observe_gradient_candidates); the GP's own state is not touched.  lml[c] is NaN
  // and grad[c] zero where status[c] == GOGP_ENOTPD.  Counterpart: candidates evaluated
  // concurrently by the reference's optimiser (optimize.Settings.Concurrent, tutorial/tutorial.go:30,141).
  struct Candidates {
    std::vector<double> lml;
    std::vector<std::vector<double>> grad;
    std::vector<int> status;
  };
  Candidates ObserveGradientCandidates(const std::vector<std::vector<double>> &xs) {
    const size_t k = xs.size(), P = ThetaSimil.size() + ThetaNoise.size();
    std::vector<double> flat(k * P), lml(k), grads(k * P);
    for (size_t c = 0; c < k; ++c) {
      if (xs[c].size() != P) throw Error(GOGP_EARG, "len(x)");
      std::copy(xs[c].begin(), xs[c].end(), flat.begin() + (long)(c * P));
    }
    Candidates out;
    out.status.assign(k, GOGP_OK);
    check(push());
    const int rc = gogp_observe_gradient_candidates(h_, (int)k, flat.data(), (int64_t)P, lml.data(), grads.data(),
                                                    out.status.data());
    if (rc != GOGP_OK && rc != GOGP_ENOTPD && rc != GOGP_ECOND) check(rc);
    out.lml = lml;
    for (size_t c = 0; c < k; ++c)
      out.grad.emplace_back(grads.begin() + (long)(c * P), grads.begin() + (long)((c + 1) * P));
    return out;
  }

  // gp.GP.L (gp/gp.go:35): lower factor, row-major n x n, fetched on demand
  std::vector<double> Factor() {
    const size_t n = (size_t)gogp_n(h_);
    std::vector<double> L(n * n);
    check(gogp_get_factor(h_, L.data()));
    return L;
  }

  // gogp_set_option (no reference counterpart), e.g. ("gradient_precision", 32): Observe's LML, Alpha and Produce stay
  // fp64, what only Gradient needs runs on the fp32 matrix cores (one-term kernels with an output scale only: refused --
  // an exception -- for sums of terms, whose scale components would come off the float K^-1 at 1.9e-4)
  void SetOption(const char *name, int64_t value) { check(gogp_set_option(h_, name, value)); }

  gogp_handle *handle() { return h_; }

 private:
  gogp_handle *h_ = nullptr;
  size_t last_len_ = 0;
  bool dirty_ = true;

  std::vector<double> pack(const std::vector<std::vector<double>> &x) const {
    std::vector<double> flat(x.size() * (size_t)NDim);
    for (size_t i = 0; i < x.size(); ++i)
      for (int d = 0; d < NDim; ++d) flat[i * NDim + d] = x[i][d];
    return flat;
  }
  int push() {
    if (!dirty_) return GOGP_OK;
    if (X.size() != Y.size()) return GOGP_EARG;
    std::vector<double> flat = pack(X);
    const int rc = gogp_set_data(h_, flat.data(), Y.data(), (int64_t)Y.size());
    if (rc == GOGP_OK) dirty_ = false;
    return rc;
  }
  int fetch_alpha() {
    Alpha.assign((size_t)gogp_n(h_), 0.0);
    return gogp_get_alpha(h_, Alpha.data());
  }
  void check(int rc) {
    if (rc != GOGP_OK) throw Error(rc, gogp_last_error(h_));
  }
};

}  // namespace gogp
