"""Priors of the reference's case studies, the host-side callers of the GP hot path through
gp.Model (gp/model.go:9-28).  The reference differentiates them with
infergo's tape; here the gradients are written out.  Log-densities as in infergo's ``dist``
package (bitbucket.org/dtolpin/infergo/dist, a dependency of the reference that is not vendored
in /root/reference): Normal.Logp(mu, sigma, x) = -((x-mu)/sigma)^2/2 - log sigma - log(2 pi)/2,
Expon.Logp(lambda, y) = log lambda - lambda y."""
import math

import numpy as np

_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def _normal_logp(mu, sigma, x):
    return -0.5 * ((x - mu) / sigma) ** 2 - math.log(sigma) - _LOG_SQRT_2PI


class HyperPriors:
    """tutorial/hyperpriors/model/model.go:9-44: normal priors on the log-parameters
    [c1, c2, l1, l2, p, s] of trend + seasonality (kernel: tutorial/hyperpriors/kernel/kernel.go);
    the seasonality weight's prior is centred log 2 below the trend weight."""

    def Observe(self, x) -> float:
        x = np.asarray(x, dtype=float)
        c1, c2, l1, l2, p, s = x[:6]
        ll = _normal_logp(-1.0, 1.0, c1)
        ll += _normal_logp(c1 - math.log(2.0), 1.0, c2)
        ll += _normal_logp(0.0, 2.0, l1) + _normal_logp(0.0, 2.0, l2)
        ll += _normal_logp(0.0, 1.0, p) + _normal_logp(0.0, 1.0, s)
        d = c2 - (c1 - math.log(2.0))
        self._grad = np.array([-(c1 + 1.0) + d, -d, -l1 / 4.0, -l2 / 4.0, -p, -s])
        return float(ll)

    def Gradient(self) -> np.ndarray:
        return self._grad


class AnyNoisePriors:
    """tutorial/anynoise/model/model.go:8-50: x = [log c, log l, log s | inputs | outputs] (the full
    Observe form, 1-D inputs).  Normal priors on the three parameters; the LATENT outputs carried
    in x are tied to the noisy outputs memorised at the first call by a Laplacian likelihood,
    Expon.Logp(1/exp(x[s]), |Y_i - x_out_i|)."""

    def __init__(self):
        self.Y = None

    def Observe(self, x) -> float:
        x = np.asarray(x, dtype=float)
        n = (x.size - 3) // 2
        out = x[3 + n:]
        if self.Y is None or len(self.Y) != n:  # first call: memoise the initial outputs
            self.Y = out.copy()
        c, l, s = x[:3]
        ll = _normal_logp(-1.0, 1.0, c) + _normal_logp(0.0, 2.0, l) + _normal_logp(-1.0, 2.0, s)
        lam = 1.0 / math.exp(s)
        dev = self.Y - out
        ll += n * math.log(lam) - lam * np.abs(dev).sum()
        g = np.zeros(x.size)
        g[0] = -(c + 1.0)
        g[1] = -l / 4.0
        g[2] = -(s + 1.0) / 4.0 - n + lam * np.abs(dev).sum()  # d/ds of n log(1/e^s) - |dev|/e^s
        g[3 + n:] = lam * np.sign(dev)                          # d/d out_i of -lam |Y_i - out_i|
        self._grad = g
        return float(ll)

    def Gradient(self) -> np.ndarray:
        return self._grad


class AnyNoiseModel:
    """tutorial/anynoise/main.go:29-44: gp.Model whose gradient w.r.t. the INPUTS is wiped (only the
    hyperparameters and the latent outputs are inferred)."""

    def __init__(self, model):
        self.Model = model
        self.GP = model.GP

    def Observe(self, x) -> float:
        return self.Model.Observe(x)

    def Gradient(self) -> np.ndarray:
        g = np.array(self.Model.Gradient(), dtype=float)
        p = self.GP._ns + self.GP._nn if hasattr(self.GP, "_ns") else self.GP._P
        n = len(self.GP.X)
        g[p:p + n * self.GP.NDim] = 0.0
        return g
