"""
viSparseGP — variational sparse GP (VFE / Titsias) with the reference's surface
(gpax/models/sparse_gp.py:24-223): learnable inducing points, SVI with Adam(b1 = 0.5), posterior
by the Woodbury route.  The per-step objective (LowRankMultivariateNormal log-density minus the
trace term) and its gradient w.r.t. the kernel parameters, the noise AND the inducing points come
from gpx_sgp_bound on the GPU; the posterior from gpx_sgp_posterior.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .. import _lib
from ..infer import dist
from ..infer.svi import fit_delta, fit_normal
from ..utils.utils import initialize_inducing_points, rng_from_key, split_in_batches
from .gp import _Progress
from .vigp import viGP


class viSparseGP(viGP):
    """
    Variational inference-based sparse Gaussian process

    Args: as viGP (kernel 'RBF' or 'Matern'; guide 'delta' or 'normal')
    """

    def __init__(self, input_dim: int, kernel: str, mean_fn: Optional[Callable] = None, kernel_prior=None,
                 mean_fn_prior=None, noise_prior=None, noise_prior_dist: Optional[dist.Distribution] = None,
                 lengthscale_prior_dist: Optional[dist.Distribution] = None, guide: str = 'delta') -> None:
        super().__init__(input_dim, kernel, mean_fn, kernel_prior, mean_fn_prior, noise_prior, noise_prior_dist,
                         lengthscale_prior_dist, guide)
        if self.kernel_name == "Periodic":
            raise NotImplementedError("viSparseGP on the MI355X path supports 'RBF' and 'Matern'")
        self.Xu = None

    def model(self, X, y=None, Xu=None, params: Optional[Dict[str, np.ndarray]] = None, **kwargs: float) -> float:
        """The reference's NumPyro program (sparse_gp.py:62-114) registers the prior sites, the factor
        `-trace_term / 2` and a LowRankMultivariateNormal likelihood with a tracer.  As ExactGP.model does for the exact
        program, this evaluates what that program defines — the log joint of the VFE objective
            log N(y | m, W W^T + noise I) - clip(tr(Kff - Qff) / noise, 0) / 2 + sum_sites log p(theta_site)
        at `params` (constrained values; default: the prior medians) and the inducing points `Xu` (default: the fitted
        self.Xu), the bound on the device (gpx_sgp_bound, no gradient).  y = None: the log prior alone.  NaN when Kuu or
        the capacitance matrix is not positive definite."""
        X = self._set_data(X)
        sites = self._sites()
        theta = {s.name: (np.full(s.shape, float(s.dist.median())) if s.shape else float(s.dist.median()))
                 for s in sites}
        if params is not None:
            theta.update({k: v for k, v in params.items() if k in theta})
        theta = self._with_deterministic(theta)
        val = 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            for s_ in sites:
                val += float(np.sum(s_.dist.log_prob(np.asarray(theta[s_.name], dtype=np.float64).reshape(-1))))
        if y is None:
            return val
        if Xu is None:
            Xu = self.Xu
        if Xu is None:
            raise ValueError("viSparseGP.model needs inducing points: pass Xu or fit the model first")
        Xu = np.asarray(Xu, dtype=np.float64)
        if Xu.ndim == 1:
            Xu = Xu[:, None]
        y = np.asarray(y, dtype=np.float64).squeeze()
        eng = _lib.get_engine(self._device)
        eng.set_train(X)
        bound, info, _ = eng.sgp_bound(self._kind, self._ell(theta), self._scalar(theta["k_scale"]),
                                       self._scalar(theta["noise"]), float(kwargs.get("jitter", 1e-6)), Xu,
                                       y - self._mean(X, theta), want_grad=False)
        return val + bound if info == 0 and np.isfinite(bound) else float("nan")

    # -- objective: bound + log prior (+ log |J|), gradient w.r.t. (u, Xu) -----------------------------
    def _sparse_log_joint(self, sites, x, Mi: int, jitter: float, jacobian: bool):
        nu = sum(s.size for s in sites)
        u, Xu = x[:nu], x[nu:].reshape(Mi, self.kernel_dim)
        theta = self._unpack(sites, u)
        eng = self._engine()
        yres = self.y_train - self._mean(self.X_train, theta)
        bound, info, g = eng.sgp_bound(self._kind, theta["k_length"], theta["k_scale"], theta["noise"], jitter, Xu,
                                       yres, want_grad=True)
        if info != 0 or not np.isfinite(bound):
            return -np.inf, np.zeros_like(x)
        val = bound
        grad = np.zeros_like(x)
        glik = {"k_length": np.asarray(g["k_length"]), "k_scale": np.array([g["k_scale"]]),
                "noise": np.array([g["noise"]])}
        off = 0
        for s in sites:
            ui = u[off:off + s.size]
            xi = s.dist.transform(ui)
            val += float(np.sum(s.dist.log_prob(xi)))
            if s.name in glik:
                gx = glik[s.name].reshape(-1)
            else:  # mean-function parameter through d bound / d yres: the full Jacobian of the site (a vector-valued
                # site — a plate inside mean_fn_prior — gets one row per element, as ExactGP's chain rule does)
                gx = -(self._dmean(self.X_train, theta, s.name) @ np.asarray(g["yres"], dtype=np.float64).reshape(-1))
            gx = gx + s.dist.grad_log_prob(xi)
            gu = gx * s.dist.dx_du(ui)
            if jacobian:
                lj, dlj = s.dist.log_abs_det_jacobian(ui)
                val += float(np.sum(lj))
                gu = gu + dlj
            grad[off:off + s.size] = gu
            off += s.size
        grad[nu:] = g["Xu"].reshape(-1)
        return val, grad

    def fit(self, rng_key, X: np.ndarray, y: np.ndarray, inducing_points_ratio: float = 0.1,
            inducing_points_selection: str = 'random', num_steps: int = 1000, step_size: float = 5e-3,
            progress_bar: bool = True, print_summary: bool = True, device=None, **kwargs: float) -> None:
        """Run variational inference to learn the sparse GP (hyper)parameters and the inducing
        points (sparse_gp.py:116-171)."""
        X, y = self._set_data(X, y)
        self._device = device if isinstance(device, int) else None
        Xu0 = np.asarray(initialize_inducing_points(X.copy(), inducing_points_ratio, inducing_points_selection, rng_key),
                         dtype=np.float64)
        self.X_train = X
        self.y_train = y
        jitter = float(kwargs.get("jitter", 1e-6))
        rng = rng_from_key(rng_key)
        sites = self._sites()
        Mi = Xu0.shape[0]
        nu = sum(s.size for s in sites)
        prog = _Progress(progress_bar, "svi")
        if self.guide_type == 'delta':
            obj = lambda x: self._sparse_log_joint(sites, x, Mi, jitter, jacobian=False)
            x, losses = fit_delta(obj, np.concatenate([self._init_delta(sites), Xu0.reshape(-1)]), num_steps, step_size,
                                  prog)
            u, scale, xu = x[:nu], None, x[nu:]
        else:
            obj = lambda x: self._sparse_log_joint(sites, x, Mi, jitter, jacobian=True)
            u, scale, losses, xu = fit_normal(obj, nu, num_steps, step_size, rng, progress=prog, point0=Xu0.reshape(-1))
        prog.close()
        self._store_guide(sites, u, scale, losses)
        self.kernel_params["Xu"] = xu.reshape(Mi, self.kernel_dim)
        self.Xu = self.kernel_params["Xu"]
        if print_summary:
            self._print_summary()

    # -- posterior (sparse_gp.py:173-223) ---------------------------------------------------------------
    def _sparse_posterior(self, X_new, params, noiseless, jitter, want_cov, want_var):
        noise = self._scalar(params["noise"])
        noise_p = noise * (1 - int(bool(noiseless)))
        y_residual = self.y_train - self._mean(self.X_train, params)
        mean, cov, var, info = self._engine().sgp_posterior(self._kind, params["k_length"],
                                                            self._scalar(params["k_scale"]), noise, jitter, self.Xu,
                                                            y_residual, X_new, noise_p, want_cov, want_var)
        if self.mean_fn is not None:
            mean = mean + self._mean(X_new, params)
        return mean, cov, var

    def get_mvn_posterior(self, X_new: np.ndarray, params: Dict[str, np.ndarray], noiseless: bool = False,
                          **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        X_new = self._set_data(X_new)
        mean, cov, _ = self._sparse_posterior(X_new, params, noiseless, float(kwargs.get("jitter", 1e-6)), True, False)
        return mean, cov

    def predict(self, rng_key, X_new: np.ndarray, samples: Optional[Dict[str, np.ndarray]] = None,
                noiseless: bool = False, device=None, **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        X_new = self._set_data(X_new)
        if isinstance(device, int):
            self._device = device
        if samples is None:
            samples = self.get_samples()
        mean, _, var = self._sparse_posterior(X_new, samples, noiseless, float(kwargs.get("jitter", 1e-6)), False, True)
        return mean, var

    def predict_in_batches(self, rng_key, X_new: np.ndarray, batch_size: int = 100,
                           samples: Optional[Dict[str, np.ndarray]] = None, predict_fn=None, noiseless: bool = False,
                           device=None, **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        X_new = self._set_data(X_new)
        if predict_fn is None:
            # mean / variance only: nothing Ms x Ms is formed, so the slice the reference needs to bound its
            # memory (sparse_gp.py predict_in_batches -> vigp.py:129-151) only costs a re-factorisation per slice
            # here.  Use device-sized slices (identical values point by point; C5: 2.2 s -> 80 ms).
            predict_fn = lambda xi: self.predict(rng_key, xi, samples, noiseless, **kwargs)
            batch_size = max(int(batch_size), 65536)
        y_pred, y_var = [], []
        for Xi in split_in_batches(X_new, batch_size, dim=0):
            m, v = predict_fn(Xi)
            y_pred.append(m)
            y_var.append(v)
        return np.concatenate(y_pred, 0), np.concatenate(y_var, 0)
