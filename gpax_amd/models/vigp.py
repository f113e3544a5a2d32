"""
viGP — variational-inference exact GP with the reference's surface (gpax/models/vigp.py:28-192):
SVI (Adam, b1 = 0.5) with a Delta (MAP) or mean-field Normal guide; predict returns
(mean, variance).  The ELBO's likelihood term and gradient are the same device lml / gradient
pass ExactGP uses; predict computes diag(cov) directly (gpx_posterior `var`) instead of forming
an M x M covariance to take its diagonal (vigp.py:184-185), and predict_in_batches factors K
once instead of re-inverting it for every slice (vigp.py:129-151).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np

from ..infer import dist
from ..infer.svi import fit_delta, fit_normal
from ..utils.utils import rng_from_key, split_in_batches
from .. import _lib
from .gp import ExactGP, _Progress


class viGP(ExactGP):
    """
    Gaussian process via stochastic variational inference

    Args: as ExactGP, plus
        guide: 'delta' (MAP, default) or 'normal' (mean-field AutoNormal)
    """

    def __init__(self, input_dim: int, kernel: str, mean_fn: Optional[Callable] = None, kernel_prior=None,
                 mean_fn_prior=None, noise_prior=None, noise_prior_dist: Optional[dist.Distribution] = None,
                 lengthscale_prior_dist: Optional[dist.Distribution] = None, guide: str = 'delta') -> None:
        super().__init__(input_dim, kernel, mean_fn, kernel_prior, mean_fn_prior, noise_prior, noise_prior_dist,
                         lengthscale_prior_dist)
        self.guide_type = 'normal' if guide == 'normal' else 'delta'
        self.svi = None
        self.kernel_params = None
        self.loss = None

    def _init_delta(self, sites):
        return np.concatenate([s.dist.inverse(np.full(s.size, s.dist.median(), dtype=np.float64)) for s in sites])

    def _run_svi(self, sites, rng, num_steps, step_size, jitter, progress_bar, extra=None):
        """Shared by viGP and viSparseGP (`extra` = additional unconstrained parameters and their
        objective hook, used for the inducing points)."""
        prog = _Progress(progress_bar, "svi")
        if self.guide_type == 'delta':
            obj = lambda u: self._log_joint(sites, u, jitter, jacobian=False)
            u, losses = fit_delta(obj, self._init_delta(sites), num_steps, step_size, prog)
            scale = None
        else:
            dim = sum(s.size for s in sites)
            obj = lambda u: self._log_joint(sites, u, jitter, jacobian=True)
            u, scale, losses = fit_normal(obj, dim, num_steps, step_size, rng, progress=prog)
        prog.close()
        return u, scale, losses

    def fit(self, rng_key, X: np.ndarray, y: np.ndarray, num_steps: int = 1000, step_size: float = 5e-3,
            progress_bar: bool = True, print_summary: bool = True, device=None, **kwargs: float) -> None:
        """Run variational inference to learn the GP (hyper)parameters (vigp.py:77-123)."""
        X, y = self._set_data(X, y)
        self._device = device if isinstance(device, int) else None
        self.X_train = X
        self.y_train = y
        jitter = float(kwargs.get("jitter", 1e-6))
        sites = self._sites()
        u, scale, losses = self._run_svi(sites, rng_from_key(rng_key), num_steps, step_size, jitter, progress_bar)
        self._store_guide(sites, u, scale, losses)
        if print_summary:
            self._print_summary()

    def _store_guide(self, sites, u, scale, losses):
        self._svi_sites = sites
        self.kernel_params = {"auto_loc": u, "auto_scale": scale}
        self.loss = losses
        self.svi = self  # the reference keeps the numpyro SVI object; `svi is not None` after fit

    def get_samples(self) -> Dict[str, np.ndarray]:
        """Guide median in the constrained space (vigp.py:125-127)."""
        if self.kernel_params is None:
            raise RuntimeError("call fit() first")
        theta = self._unpack(self._svi_sites, self.kernel_params["auto_loc"])
        return {k: np.asarray(v) for k, v in theta.items()}

    def predict_in_batches(self, rng_key, X_new: np.ndarray, batch_size: int = 100,
                           samples: Optional[Dict[str, np.ndarray]] = None, predict_fn=None, noiseless: bool = False,
                           device=None, **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        """predict() over slices of X_new (vigp.py:129-151).  K is factored once for all slices.
        `device`: a GPU ordinal, or "all" / a list of ordinals: the slices are dealt in contiguous blocks over those
        GPUs (one host thread and one context each; every GPU factors K(theta) itself — 30 ms at N = 16384, less than
        moving the 2 GB factor over xGMI).  Slice by slice the values are those of one GPU."""
        X_new = self._set_data(X_new)
        if isinstance(device, int):
            self._device = device
        if samples is None:
            samples = self.get_samples()
        jitter = float(kwargs.get("jitter", 1e-6))
        devs = [None]
        if predict_fn is None:
            # mean and variance of a test point do not depend on which other points share its slice (row-wise solves,
            # and the same fma chains in every tile shape): the slices the reference needs to bound its M x M covariance
            # (vigp.py:129-151) only cost launches here.  Device-sized slices: up to ~4 GB of k_pX rows at a time, and at
            # least one slice per GPU when several are asked for — identical values point by point (C5: 263 slices of
            # 1000 pixels -> 17 of 16384).
            devs = self._slice_devices(device, X_new.shape[0])
            rows_fit = max(1, int(4.0e9 / (8.0 * max(1, self.X_train.shape[0]))))
            chunk = min(16384, rows_fit)
            if len(devs) > 1:
                chunk = min(chunk, -(-X_new.shape[0] // len(devs)))
            batch_size = max(int(batch_size), chunk) if len(devs) == 1 else max(1, chunk)
        slices = split_in_batches(X_new, batch_size, dim=0)
        if len(devs) > len(slices):
            devs = devs[:len(slices)] if len(slices) > 1 else [None]
        if len(devs) > 1:
            y_pred, y_var, info = self._predict_slices_on(devs, slices, samples, noiseless, jitter)
        else:
            info = 0
            if predict_fn is None:
                _, info = self._factor_at(samples, jitter)
                predict_fn = lambda xi: self._posterior_mean_var(xi, samples, noiseless, jitter)
            y_pred, y_var = [], []
            for Xi in slices:
                m, v = predict_fn(Xi)
                y_pred.append(m)
                y_var.append(v)
        y_pred, y_var = np.concatenate(y_pred, 0), np.concatenate(y_var, 0)
        if info != 0:  # K(theta) not positive definite: NaN like predict() and the reference's inverse route
            y_pred, y_var = np.full_like(y_pred, np.nan), np.full_like(y_var, np.nan)
        return y_pred, y_var

    def _slice_devices(self, device, nslices: int):
        if not isinstance(_lib.get_engine(self._device), _lib.Engine):
            return [None]  # an injected engine (CPU tests)
        if device == "all":
            devs = list(range(_lib.visible_device_count()))
        elif isinstance(device, (list, tuple)):
            devs = [int(v) for v in device]
        else:
            return [None]
        devs = devs[:max(1, nslices)]
        return devs if len(devs) > 1 else [None]

    def _predict_slices_on(self, devs, slices, samples, noiseless, jitter):
        """Contiguous blocks of the slices on the GPUs `devs`: per GPU a context of its own, X_train uploaded, K(theta)
        factored once, then its slices one after the other (vigp.py:129-151 re-inverts K for every slice)."""
        import threading
        G = len(devs)
        bounds = [_lib.shard_range(len(slices), g, G) for g in range(G)]
        out_m, out_v = [None] * len(slices), [None] * len(slices)
        infos, errors = [0] * G, []

        def work(g):
            eng = None
            try:
                eng = self._engine_on(devs[g])
                _, infos[g] = self._factor_at(samples, jitter, eng=eng)
                for i in range(*bounds[g]):
                    out_m[i], out_v[i] = self._posterior_mean_var(slices[i], samples, noiseless, jitter, eng=eng)
            except Exception as ex:  # surface worker failures in the caller
                errors.append(ex)
            finally:
                if eng is not None:
                    eng.close()

        ts = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errors:
            raise errors[0]
        return out_m, out_v, max(infos, key=abs)

    def _factor_at(self, params, jitter, eng=None):
        noise = self._scalar(params["noise"])
        y_residual = self.y_train - self._mean(self.X_train, params)
        eng = self._engine() if eng is None else eng
        return eng.factor(self._kind, self._ell(params), self._scalar(params["k_scale"]), noise, jitter, y_residual)

    def _posterior_mean_var(self, X_new, params, noiseless, jitter, eng=None):
        noise_p = self._scalar(params["noise"]) * (1 - int(bool(noiseless)))
        eng = self._engine() if eng is None else eng
        mean, _, var = eng.posterior(X_new, noise_p, jitter, want_cov=False, want_var=True)
        if self.mean_fn is not None:
            mean = mean + self._mean(X_new, params)
        return mean, var

    def predict(self, rng_key, X_new: np.ndarray, samples: Optional[Dict[str, np.ndarray]] = None,
                noiseless: bool = False, device=None, **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        """Posterior mean and variance at X_new (vigp.py:153-185)."""
        X_new = self._set_data(X_new)
        if isinstance(device, int):
            self._device = device
        if samples is None:
            samples = self.get_samples()
        jitter = float(kwargs.get("jitter", 1e-6))
        lml, info = self._factor_at(samples, jitter)
        mean, var = self._posterior_mean_var(X_new, samples, noiseless, jitter)
        if info != 0:
            mean, var = np.full_like(mean, np.nan), np.full_like(var, np.nan)
        return mean, var

    def _print_summary(self) -> None:
        params_map = self.get_samples()
        print('\nInferred GP parameters')
        for (k, vals) in params_map.items():
            spaces = " " * (15 - len(k))
            print(k, spaces, np.around(vals, 4))
