"""
Minimal prior distributions for the exact-GP hyperparameters.

The reference takes `numpyro.distributions` objects for `noise_prior_dist` /
`lengthscale_prior_dist` and defaults to LogNormal(0, 1) (gpax/models/gp.py:222-247); the helpers
in gpax/priors/priors.py:71-189 hand out Normal, LogNormal, HalfNormal, Gamma and Uniform.
NumPyro is not a dependency here, so the five are restated: log-density, its derivative, the
support transform to the unconstrained space NUTS/SVI work in (NumPyro's biject_to: exp for
positive supports, affine sigmoid for an interval, identity for the real line) and sampling.
"""
from __future__ import annotations

import math

import numpy as np
from scipy import special

__all__ = ["Distribution", "Normal", "LogNormal", "HalfNormal", "HalfCauchy", "Gamma", "Uniform"]


class Distribution:
    support = "real"  # "real" | "positive" | "interval"

    # -- density in the constrained space ------------------------------------------------------
    def log_prob(self, x):
        raise NotImplementedError

    def grad_log_prob(self, x):
        raise NotImplementedError

    def sample(self, rng: np.random.Generator, shape=()):
        raise NotImplementedError

    def median(self):
        raise NotImplementedError

    # -- support transform: x = T(u), u unconstrained ---------------------------------------------
    def transform(self, u):
        if self.support == "real":
            return u
        if self.support == "positive":
            return np.exp(u)
        lo, hi = self._bounds()
        return lo + (hi - lo) * special.expit(u)

    def inverse(self, x):
        if self.support == "real":
            return x
        if self.support == "positive":
            return np.log(x)
        lo, hi = self._bounds()
        return special.logit((x - lo) / (hi - lo))

    def dx_du(self, u):
        if self.support == "real":
            return np.ones_like(u)
        if self.support == "positive":
            return np.exp(u)
        lo, hi = self._bounds()
        s = special.expit(u)
        return (hi - lo) * s * (1 - s)

    def log_abs_det_jacobian(self, u):
        """log |dx/du| and its derivative w.r.t. u."""
        if self.support == "real":
            return np.zeros_like(u), np.zeros_like(u)
        if self.support == "positive":
            return u, np.ones_like(u)
        lo, hi = self._bounds()
        s = special.expit(u)
        return math.log(hi - lo) + np.log(s) + np.log1p(-s), 1 - 2 * s

    def _bounds(self):
        raise NotImplementedError


class Normal(Distribution):
    support = "real"

    def __init__(self, loc=0.0, scale=1.0):
        self.loc, self.scale = float(loc), float(scale)

    def log_prob(self, x):
        z = (x - self.loc) / self.scale
        return -0.5 * z * z - math.log(self.scale) - 0.5 * math.log(2 * math.pi)

    def grad_log_prob(self, x):
        return -(x - self.loc) / self.scale ** 2

    def sample(self, rng, shape=()):
        return self.loc + self.scale * rng.standard_normal(shape)

    def median(self):
        return self.loc


class LogNormal(Distribution):
    support = "positive"

    def __init__(self, loc=0.0, scale=1.0):
        self.loc, self.scale = float(loc), float(scale)

    def log_prob(self, x):
        lx = np.log(x)  # callers on the sampling path wrap this in np.errstate (x = exp(u) can underflow to 0)
        z = (lx - self.loc) / self.scale
        return -0.5 * z * z - math.log(self.scale) - 0.5 * math.log(2 * math.pi) - lx

    def grad_log_prob(self, x):
        return (-(np.log(x) - self.loc) / self.scale ** 2 - 1.0) / x

    def sample(self, rng, shape=()):
        return np.exp(self.loc + self.scale * rng.standard_normal(shape))

    def median(self):
        return math.exp(self.loc)


class HalfNormal(Distribution):
    support = "positive"

    def __init__(self, scale=1.0):
        self.scale = float(scale)

    def log_prob(self, x):
        z = x / self.scale
        return -0.5 * z * z + 0.5 * math.log(2 / math.pi) - math.log(self.scale)

    def grad_log_prob(self, x):
        return -x / self.scale ** 2

    def sample(self, rng, shape=()):
        return np.abs(self.scale * rng.standard_normal(shape))

    def median(self):
        return self.scale * 0.6744897501960817


class HalfCauchy(Distribution):
    """Prior of the LinReg noise scale (gpax/models/linreg.py:27)."""
    support = "positive"

    def __init__(self, scale=1.0):
        self.scale = float(scale)

    def log_prob(self, x):
        z = x / self.scale
        return math.log(2.0 / (math.pi * self.scale)) - np.log1p(z * z)

    def grad_log_prob(self, x):
        return -2.0 * x / (self.scale ** 2 + x * x)

    def sample(self, rng, shape=()):
        return np.abs(self.scale * np.tan(math.pi * (rng.uniform(size=shape) - 0.5)))

    def median(self):
        return self.scale


class Gamma(Distribution):
    support = "positive"

    def __init__(self, concentration, rate=1.0):
        self.concentration, self.rate = float(concentration), float(rate)

    def log_prob(self, x):
        a, b = self.concentration, self.rate
        return a * math.log(b) - special.gammaln(a) + (a - 1) * np.log(x) - b * x

    def grad_log_prob(self, x):
        return (self.concentration - 1) / x - self.rate

    def sample(self, rng, shape=()):
        return rng.gamma(self.concentration, 1.0 / self.rate, shape)

    def median(self):
        return float(special.gammaincinv(self.concentration, 0.5) / self.rate)


class Uniform(Distribution):
    support = "interval"

    def __init__(self, low=0.0, high=1.0):
        self.low, self.high = float(low), float(high)

    def _bounds(self):
        return self.low, self.high

    def log_prob(self, x):
        return np.zeros_like(np.asarray(x, dtype=np.float64)) - math.log(self.high - self.low)

    def grad_log_prob(self, x):
        return np.zeros_like(np.asarray(x, dtype=np.float64))

    def sample(self, rng, shape=()):
        return rng.uniform(self.low, self.high, shape)

    def median(self):
        return 0.5 * (self.low + self.high)
