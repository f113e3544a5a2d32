"""
MeasuredNoiseGP — fully Bayesian GP that takes measured (per-point) noise variances instead of inferring a
noise level, with the reference's surface (gpax/models/mngp.py:30-256).

Fit: the training covariance is kernel(X, X, theta, 0, jitter) + diag(measured_noise) (mngp.py:92-98);
on the device that is the per-point diagonal vector of gpx_set_diag, everything else is ExactGP's fit step.
Predict (mngp.py:159-246): the posterior for each HMC sample is ExactGP.get_mvn_posterior with the
deterministic site noise = 0 — i.e. the training block is kernel + jitter * I, WITHOUT the measured noise (the
reference does not override get_mvn_posterior; mirrored as is) — then the noise variance extrapolated to
X_new ('linreg' / 'gpreg') is added to diag(cov) and samples are drawn from the MARGINALS only.  That needs
means and variances, not covariances: one gpx_predict_sweep with n = 0 and the variance output.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np

from ..infer import dist
from ..utils.utils import get_keys, rng_from_key
from .gp import ExactGP, _Site
from .linreg import LinReg
from .vigp import viGP


class MeasuredNoiseGP(ExactGP):
    """
    Gaussian Process model that incorporates measured noise.

    Args:
        input_dim, kernel, mean_fn, kernel_prior, mean_fn_prior, lengthscale_prior_dist: as ExactGP
    """

    def __init__(self, input_dim: int, kernel: str, mean_fn: Optional[Callable] = None, kernel_prior=None,
                 mean_fn_prior=None, lengthscale_prior_dist: Optional[dist.Distribution] = None) -> None:
        super().__init__(input_dim, kernel, mean_fn, kernel_prior, mean_fn_prior, None, None, lengthscale_prior_dist)
        self.measured_noise = None
        self.noise_predicted = None
        self._use_measured = False
        self._noise_version = 0

    # -- model (mngp.py:74-98): no noise site; noise is the deterministic 0 --------------------------------
    def _sites(self):
        return [s for s in super()._sites() if s.name != "noise"]

    def _unpack(self, sites, u):
        theta = super()._unpack(sites, u)
        theta["noise"] = 0.0
        return theta

    def model(self, X, y=None, measured_noise=None, params: Optional[Dict[str, np.ndarray]] = None, **kwargs) -> float:
        """What the reference's NumPyro program defines (mngp.py:74-98), evaluated as ExactGP.model evaluates the exact one:
        sum of the site log-densities + log N(y | m, k(theta) + jitter I + diag(measured_noise)) at `params` (default: the
        prior medians) — there is no noise site, noise is the deterministic 0.  y = None: the log prior alone."""
        from .. import _lib
        X = self._set_data(X)
        theta, val = self._theta_and_log_prior(self._sites(), params)
        theta["noise"] = 0.0
        if y is None:
            return val
        if measured_noise is None:
            raise ValueError("MeasuredNoiseGP.model needs the measured noise variances of the training points")
        mn = np.ascontiguousarray(np.asarray(measured_noise, dtype=np.float64).reshape(-1))
        if mn.shape[0] != X.shape[0]:
            raise ValueError("measured_noise must have one value per training point")
        y = np.asarray(y, dtype=np.float64).squeeze()
        eng = _lib.get_engine(self._device)
        eng.set_train(X)
        eng.set_diag(mn)
        try:  # the per-point diagonal must not outlive this call on the shared context
            lml, info = eng.factor(self._kind, self._ell(theta), self._scalar(theta["k_scale"]), 0.0,
                                   float(kwargs.get("jitter", 1e-6)), y - self._mean(X, theta))
        finally:
            eng.set_diag(None)
            eng._diag_key = None
        return val + lml if info == 0 else float("nan")

    def _engine(self):
        eng = super()._engine()
        want = self.measured_noise if self._use_measured else None
        # (re)assert the per-point diagonal of this model's current phase on the shared context
        # keyed on the fit counter, not on object identities: a second fit() on the same X array with other
        # measured noise must upload the new vector (Engine.set_train also clears the key)
        key = (id(self), self._data_version, self._noise_version, want is not None)
        if getattr(eng, "_diag_key", None) != key:
            eng.set_diag(want)
            eng._diag_key = key
        return eng

    def _prepare_engine(self, eng) -> None:
        super()._prepare_engine(eng)
        eng.set_diag(self.measured_noise if self._use_measured else None)

    def fit(self, rng_key, X: np.ndarray, y: np.ndarray, measured_noise: np.ndarray, num_warmup: int = 2000,
            num_samples: int = 2000, num_chains: int = 1, chain_method: str = "sequential", progress_bar: bool = True,
            print_summary: bool = True, device=None, **kwargs: float) -> None:
        """Run HMC to infer the GP parameters; measured_noise: 1D vector of measured noise variances."""
        self.measured_noise = np.ascontiguousarray(np.asarray(measured_noise, dtype=np.float64).reshape(-1))
        if self.measured_noise.shape[0] != self._set_data(X).shape[0]:
            raise ValueError("measured_noise must have one value per training point")
        self.noise_predicted = None
        self._noise_version += 1
        self._use_measured = True
        try:
            super().fit(rng_key, X, y, num_warmup, num_samples, num_chains, chain_method, progress_bar, False, device,
                        **kwargs)
        finally:
            self._use_measured = False
        # numpyro.deterministic("noise", 0.0) shows up among the samples (mngp.py:84)
        self._samples["noise"] = np.zeros(self._chain_shape)
        if print_summary:
            self._print_summary()

    # -- prediction (mngp.py:159-246) ----------------------------------------------------------------------
    def _marginal_draws(self, rng, means, vars_, noise_predicted, n):
        """y_sampled[s, i] = mean_s + sqrt(clip(diag(K_s) + noise_predicted, 0)) * N(0, 1)  (mngp.py:170-181)"""
        sig = np.sqrt(np.clip(vars_ + noise_predicted[None, :], 0.0, None))
        eps = rng.standard_normal((means.shape[0], n, means.shape[1]))
        return means[:, None, :] + sig[:, None, :] * eps

    def _predict(self, rng_key, X_new: np.ndarray, params: Dict[str, np.ndarray], noise_predicted: np.ndarray, n: int,
                 noiseless: bool = False, **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        y_mean, K = self.get_mvn_posterior(X_new, params, noiseless, **kwargs)
        var = np.diag(K)
        y_sampled = self._marginal_draws(rng_from_key(rng_key), y_mean[None], var[None],
                                         np.asarray(noise_predicted, dtype=np.float64), n)[0]
        return y_mean, y_sampled

    def predict(self, rng_key, X_new: np.ndarray, samples: Optional[Dict[str, np.ndarray]] = None, n: int = 1,
                filter_nans: bool = False, noiseless: bool = True, device=None,
                noise_prediction_method: str = 'linreg', **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        """Returns the centre of mass of the sampled means (M,) and all sampled predictions (S, n, M)."""
        if noise_prediction_method not in ["linreg", "gpreg"]:
            raise NotImplementedError("For noise prediction method, select between 'linreg' and 'gpreg'")
        noise_pred_fn = self.linreg if noise_prediction_method == "linreg" else self.gpreg
        X_new = self._set_data(X_new)
        if self.noise_predicted is not None:  # cached by the first call, like the reference (mngp.py:228-232)
            noise_predicted = self.noise_predicted
        else:
            noise_predicted = np.asarray(noise_pred_fn(self.X_train, self.measured_noise, X_new, **kwargs),
                                         dtype=np.float64).reshape(-1)
            self.noise_predicted = noise_predicted
        if samples is None:
            samples = self.get_samples(chain_dim=False)
        if isinstance(device, int):
            self._device = device
        jitter = float(kwargs.get("jitter", 1e-6))
        S = len(next(iter(samples.values())))
        d = self.kernel_dim
        ells = np.asarray(samples["k_length"], dtype=np.float64).reshape(S, -1)
        ells = np.ascontiguousarray(np.broadcast_to(ells, (S, d)))
        if self.kernel_name == "Periodic":
            ells = np.ascontiguousarray(np.concatenate(
                [ells, np.asarray(samples["period"], dtype=np.float64).reshape(S, 1)], axis=1))
        scales = np.asarray(samples["k_scale"], dtype=np.float64).reshape(S)
        noises = np.asarray(samples.get("noise", np.zeros(S)), dtype=np.float64).reshape(S)
        mean_shift = None
        if self.mean_fn is not None:
            per = [{k: np.asarray(v)[s] for k, v in samples.items()} for s in range(S)]
            yres = np.stack([self.y_train - self._mean(self.X_train, p) for p in per])
            mean_shift = np.stack([self._mean(X_new, p) for p in per])
        else:
            yres = self.y_train
        eng = self._engine()  # predict phase: no per-point diagonal on the training block
        means, _, infos, vars_ = eng.predict_sweep(self._kind, ells, scales, noises, yres, X_new, noiseless, jitter,
                                                   None, want_var=True)
        if mean_shift is not None:
            means = means + mean_shift
        y_sampled = self._marginal_draws(rng_from_key(rng_key), means, vars_, noise_predicted, n)
        if filter_nans:
            keep = ~np.isnan(y_sampled).any(axis=(1, 2))
            y_sampled = y_sampled[keep]
        return means.mean(0), y_sampled

    def linreg(self, x, y, x_new, **kwargs):
        lreg = LinReg()
        lreg.train(x, y, **{k: v for k, v in kwargs.items() if k in ("learning_rate", "num_iterations")})
        return lreg.predict(x_new)

    def gpreg(self, x, y, x_new, **kwargs):
        keys = get_keys()
        vigp = viGP(self.kernel_dim, 'RBF')
        fit_kw = {k: v for k, v in kwargs.items() if k in ("num_steps", "step_size", "jitter")}
        vigp.fit(keys[0], x, y, progress_bar=False, print_summary=False, device=self._device, **fit_kw)
        return vigp.predict(keys[1], x_new, noiseless=True, device=self._device)[0]
