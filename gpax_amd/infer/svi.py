"""
Host-side stochastic variational inference loops (Adam) for viGP / viSparseGP.

Replaces numpyro.infer.SVI + Trace_ELBO + AutoDelta / AutoNormal + numpyro.optim.Adam as used in
gpax/models/vigp.py:108-120 (Adam(step_size, b1=0.5), one ELBO particle).
  * guide='delta'  : MAP in the constrained space, optimised through the support transform —
                     objective log p(y, theta(u)) with NO Jacobian term (a Delta guide has zero
                     log-density), initialised at the prior medians (NumPyro's init_to_median is a
                     15-draw sample median; here the exact median, so runs are deterministic).
  * guide='normal' : mean-field Normal in the unconstrained space, reparameterised gradient,
                     loc ~ U(-2, 2) (init_to_uniform), scale = softplus(rho) initialised to 0.1.
"""
from __future__ import annotations

import math
from typing import Callable, Tuple

import numpy as np


class Adam:
    def __init__(self, dim: int, step_size: float = 5e-3, b1: float = 0.5, b2: float = 0.999, eps: float = 1e-8):
        self.step_size, self.b1, self.b2, self.eps = step_size, b1, b2, eps
        self.m = np.zeros(dim)
        self.v = np.zeros(dim)
        self.t = 0

    def step(self, x: np.ndarray, grad_loss: np.ndarray) -> np.ndarray:
        self.t += 1
        self.m = self.b1 * self.m + (1 - self.b1) * grad_loss
        self.v = self.b2 * self.v + (1 - self.b2) * grad_loss * grad_loss
        mhat = self.m / (1 - self.b1 ** self.t)
        vhat = self.v / (1 - self.b2 ** self.t)
        return x - self.step_size * mhat / (np.sqrt(vhat) + self.eps)


def _softplus(x):
    return np.logaddexp(0.0, x)


def _inv_softplus(y):
    return y + np.log(-np.expm1(-y))


def fit_delta(objective_and_grad: Callable[[np.ndarray], Tuple[float, np.ndarray]], u0: np.ndarray, num_steps: int,
              step_size: float, progress=None):
    """Maximise objective(u) (= log joint at theta(u), no Jacobian).  Returns (u, losses)."""
    u = np.array(u0, dtype=np.float64)
    opt = Adam(u.size, step_size)
    losses = np.empty(num_steps)
    for it in range(num_steps):
        f, g = objective_and_grad(u)
        if not np.isfinite(f):
            losses[it] = np.nan  # NumPyro keeps going with NaN losses; we simply skip the update
        else:
            losses[it] = -f
            u = opt.step(u, -g)
        if progress is not None:
            progress(it, num_steps, dict(loss=losses[it]))
    return u, losses


def fit_normal(logdensity_and_grad: Callable[[np.ndarray], Tuple[float, np.ndarray]], dim: int, num_steps: int,
               step_size: float, rng: np.random.Generator, init_scale: float = 0.1, progress=None,
               point0: np.ndarray = None, b1: float = 0.5):
    """Mean-field Normal guide on the unconstrained density (log joint + log|J|) of the first `dim`
    variables; `point0` are additional point-estimated parameters (numpyro.param, e.g. the inducing
    points) appended to the argument of `logdensity_and_grad`.  Returns (loc, scale, losses[, point])."""
    npnt = 0 if point0 is None else int(np.size(point0))
    loc = rng.uniform(-2.0, 2.0, dim)
    rho = np.full(dim, _inv_softplus(init_scale))
    x = np.concatenate([loc, rho] + ([np.ravel(point0).astype(np.float64)] if npnt else []))
    opt = Adam(2 * dim + npnt, step_size, b1=b1)
    losses = np.empty(num_steps)
    for it in range(num_steps):
        loc, rho, pnt = x[:dim], x[dim:2 * dim], x[2 * dim:]
        sigma = _softplus(rho)
        e = rng.standard_normal(dim)
        u = loc + sigma * e
        f, g = logdensity_and_grad(np.concatenate([u, pnt]) if npnt else u)
        if not np.isfinite(f):
            losses[it] = np.nan
        else:
            entropy = np.sum(np.log(sigma)) + 0.5 * dim * (1 + math.log(2 * math.pi))
            losses[it] = -(f + entropy)
            g_loc = g[:dim]
            g_sigma = g[:dim] * e + 1.0 / sigma
            g_rho = g_sigma / (1.0 + np.exp(-rho))
            x = opt.step(x, -np.concatenate([g_loc, g_rho, g[dim:]]))
        if progress is not None:
            progress(it, num_steps, dict(loss=losses[it]))
    if npnt:
        return x[:dim].copy(), _softplus(x[dim:2 * dim]), losses, x[2 * dim:].copy()
    return x[:dim].copy(), _softplus(x[dim:]), losses
