"""
LinReg — the small Bayesian linear regression MeasuredNoiseGP uses to extrapolate measured noise
variances to new inputs (gpax/models/linreg.py:18-56): beta ~ Normal(0, 10), alpha ~ Normal(0, 10),
sigma ~ HalfCauchy(1), obs ~ Normal(alpha + x beta, sigma); SVI with a diagonal-Normal guide, Adam(0.01),
5000 steps; prediction at the guide median.  Host arithmetic only (O(N d) per step).
"""
from __future__ import annotations

import numpy as np

from ..infer import dist
from ..infer.svi import fit_normal


class LinReg:
    """Simple linear regression model"""

    def __init__(self):
        self.params = None

    def train(self, x, y, learning_rate: float = 0.01, num_iterations: int = 5000, **kwargs):
        x = np.asarray(x, dtype=np.float64)
        x = x if x.ndim > 1 else x[:, None]
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        n, d = x.shape
        p_beta, p_alpha, p_sigma = dist.Normal(0.0, 10.0), dist.Normal(0.0, 10.0), dist.HalfCauchy(1.0)

        def logdensity(u):  # u = [beta (d), alpha, log sigma]; log joint + log |d sigma / d u|
            beta, alpha, us = u[:d], u[d], u[d + 1:]
            sigma = float(p_sigma.transform(us)[0])
            r = y - alpha - x @ beta
            val = (-0.5 * float(r @ r) / sigma ** 2 - n * np.log(sigma) - 0.5 * n * np.log(2 * np.pi)
                   + float(np.sum(p_beta.log_prob(beta))) + float(p_alpha.log_prob(alpha))
                   + float(p_sigma.log_prob(sigma)))
            lj, dlj = p_sigma.log_abs_det_jacobian(us)
            val += float(lj[0])
            g = np.empty_like(u)
            g[:d] = x.T @ r / sigma ** 2 + p_beta.grad_log_prob(beta)
            g[d] = np.sum(r) / sigma ** 2 + p_alpha.grad_log_prob(alpha)
            g_sigma = float(r @ r) / sigma ** 3 - n / sigma + p_sigma.grad_log_prob(sigma)
            g[d + 1] = g_sigma * float(p_sigma.dx_du(us)[0]) + float(dlj[0])
            return val, g

        rng = np.random.default_rng(0)  # the reference fixes jax.random.PRNGKey(0) here (linreg.py:40)
        loc, scale, losses = fit_normal(logdensity, d + 2, int(num_iterations), float(learning_rate), rng, b1=0.9)
        self.params = {"beta": loc[:d].copy(), "alpha": float(loc[d]), "sigma": float(np.exp(loc[d + 1]))}
        self.loss = losses

    def predict(self, x_new):
        x_new = np.asarray(x_new, dtype=np.float64)
        x_new = x_new if x_new.ndim > 1 else x_new[:, None]
        return self.params["alpha"] + x_new @ self.params["beta"]

    def get_params(self):
        return self.params
