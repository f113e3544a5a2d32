"""
vExactGP — fully Bayesian GP for vector-valued targets with the reference's surface
(gpax/models/vgp.py:23-208): T independent exact GPs ("tasks"), each with its own inputs X[t], targets
y[t] and kernel hyper-parameters, inferred jointly with NUTS.

The reference vmaps the kernel and the MVN over the task axis (vgp.py:87-96,173-176).  Here the task is a
grid dimension of the same batched launches the predictive sweep uses: one `gpx_fit_batch` evaluates the T
log-likelihoods and gradients of a leapfrog, one `gpx_predict_sweep` the S x T posteriors of predict
(entry b of a batch is task b % T, reading X[t] / X_new[t] / y[t] through per-task strides).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from .. import _lib
from ..infer import dist
from ..utils.utils import rng_from_key
from .gp import ExactGP, _Site


class vExactGP(ExactGP):
    """
    Gaussian process class for vector-valued targets

    Args: as ExactGP.  X: (T, N) or (T, N, d); y: (T, N).
    """

    # -- data plumbing (vgp.py:199-208) -----------------------------------------------------------------
    def _set_data(self, X, y=None):
        X = np.asarray(X, dtype=np.float64)
        X = X[..., None] if X.ndim == 2 else X  # add feature pseudo-dimension
        X = np.ascontiguousarray(X)
        if y is not None:
            y = np.ascontiguousarray(np.asarray(y, dtype=np.float64))
            if y.shape[0] != X.shape[0]:
                raise AssertionError("Task dimensions must be identical in inputs and targets")
            return X, y
        return X

    def _set_training_data(self, X_train_new=None, y_train_new=None, device=None) -> None:
        if X_train_new is not None:
            self.X_train = self._set_data(X_train_new)
        if y_train_new is not None:
            self.y_train = np.ascontiguousarray(np.asarray(y_train_new, dtype=np.float64))
        if isinstance(device, int):
            self._device = device

    @property
    def _tasks(self) -> int:
        return self.X_train.shape[0]

    def model(self, X, y=None, params: Optional[Dict[str, np.ndarray]] = None, **kwargs: float) -> float:
        """What the reference's NumPyro program defines (vgp.py:62-96), evaluated as ExactGP.model evaluates the exact one:
        the site log-densities (per-task plates) + the sum over the T tasks of log N(y_t | m_t, k_t + (noise_t + jitter) I)
        at `params` (default: the prior medians).  X (T, N, d), y (T, N); y = None: the log prior alone."""
        jitter = float(kwargs.get("jitter", 1e-6))
        if y is None:
            X = self._set_data(X)
            yy = np.zeros(X.shape[:2])
        else:
            X, yy = self._set_data(X, y)
        with self._TrainingData(self, X, yy):
            sites = self._sites()
            theta, val = self._theta_and_log_prior(sites, params)
            if y is None:
                return val
            v, _ = self._log_joint(sites, self._unconstrained(sites, theta), jitter, jacobian=False)
            return float(v) if np.isfinite(v) else float("nan")

    # -- sample sites (vgp.py:98-120) -------------------------------------------------------------------
    def _sites(self):
        T = self._tasks
        sites = self._task_kernel_sites(T) + self._task_noise_sites(T)
        for name, d in self._mean_prior_dict().items():
            sites.append(_Site(name, (), d))
        return sites

    def _task_kernel_sites(self, T):
        """vgp.py:69-72: a `kernel_prior` callable replaces the default sites; the kernel is vmapped over the task axis
        (vgp.py:84), so every value it returns must carry the T tasks on its first axis."""
        if self.kernel_prior is not None:
            self._det = {}
            sites, returned, det = self._traced(self.kernel_prior, "kernel_prior")
            have = set(returned) if isinstance(returned, dict) else set()
            need = {"k_length", "k_scale"} | ({"period"} if self.kernel_name == "Periodic" else set())
            if not need <= have:
                raise ValueError(f"kernel_prior must return {sorted(need)} (kernels.py:44-117)")
            if det:
                raise NotImplementedError("vExactGP kernel_prior: deterministic sites have no per-task MI355X path")
            for sx in sites:
                ok = {"k_length": [(T, self.kernel_dim), (T, 1), (T,)]}.get(sx.name, [(T,), (T, 1)])
                if tuple(sx.shape) not in ok:
                    raise ValueError(f"vExactGP kernel_prior: site {sx.name!r} has shape {tuple(sx.shape)}; with {T} tasks "
                                     f"it must be one of {ok} (the kernel is vmapped over the task axis, vgp.py:84)")
            return sites
        # the default priors are the reference's program (vgp.py:101-120), traced — a subclass may override it
        sites, _, det = self._traced(lambda: self._sample_kernel_params(task_dim=T), "_sample_kernel_params")
        if det:
            raise NotImplementedError("vExactGP: deterministic kernel sites have no per-task MI355X path")
        return sites

    def _sample_kernel_params(self, task_dim: int = None) -> Dict[str, np.ndarray]:
        """The reference's default kernel priors as a program (vgp.py:101-120): k_length LogNormal(0, 1) under the plates
        (task, dim -2) x (lengthscale, dim -1); k_scale — which the reference hands `lengthscale_prior_dist` to, mirrored
        as is — and period under a task plate."""
        from ..infer.primitives import plate, sample
        length_dist = self.lengthscale_prior_dist if self.lengthscale_prior_dist is not None else dist.LogNormal(0.0, 1.0)
        with plate("plate_1", task_dim, dim=-2):  # task dimension
            with plate("lengthscale", self.kernel_dim, dim=-1):
                length = sample("k_length", dist.LogNormal(0.0, 1.0))
        period = None
        with plate("plate_2", task_dim):
            scale = sample("k_scale", length_dist)
            if self.kernel_name == "Periodic":
                period = sample("period", dist.LogNormal(0.0, 1.0))
        return {"k_length": length, "k_scale": scale, "period": period}

    def _sample_noise(self, task_dim: int = None):
        """sample("noise", noise_prior_dist or LogNormal(0, 1)) under plate "noise_plate" (vgp.py:89-99)."""
        from ..infer.primitives import plate, sample
        noise_dist = self.noise_prior_dist if self.noise_prior_dist is not None else dist.LogNormal(0.0, 1.0)
        with plate("noise_plate", task_dim):
            return sample("noise", noise_dist)

    def _task_noise_sites(self, T):
        if self.noise_prior is not None:  # vgp.py:74-75
            sites, _, _ = self._traced(self.noise_prior, "noise_prior")
        else:
            sites, _, _ = self._traced(lambda: self._sample_noise(T), "_sample_noise")
        if len(sites) != 1 or sites[0].name != "noise" or tuple(sites[0].shape) not in [(T,), (T, 1)]:
            raise ValueError(f"vExactGP noise prior must sample exactly one site 'noise' of shape ({T},): one "
                             "noise variance per task (vgp.py:74-75, 84)")
        return sites

    def _ells(self, theta) -> np.ndarray:
        """(T, n_ell): per-task lengthscales (+ period)."""
        T, d = self._tasks, self.kernel_dim
        ell = np.broadcast_to(np.asarray(theta["k_length"], dtype=np.float64).reshape(T, -1), (T, d))
        if self.kernel_name == "Periodic":
            ell = np.concatenate([ell, np.asarray(theta["period"], dtype=np.float64).reshape(T, 1)], axis=1)
        return np.ascontiguousarray(ell)

    def _residual(self, theta) -> np.ndarray:
        if self.mean_fn is None:
            return self.y_train
        return self.y_train - self._mean(self.X_train, theta)

    # -- log joint (vgp.py:62-96): sum of the T task log-likelihoods + priors ---------------------------
    def _log_joint(self, sites, u, jitter: float, jacobian: bool, want_grad: bool = True, eng=None):
        return self._log_joint_batch(sites, [u], jitter, jacobian, eng=eng)[0]

    def _log_joint_batch(self, sites, us, jitter: float, jacobian: bool, eng=None):
        if eng is None:
            eng = self._engine()
        T = self._tasks
        thetas = [self._unpack(sites, u) for u in us]
        ells = np.concatenate([self._ells(t) for t in thetas])
        scales = np.concatenate([np.asarray(t["k_scale"], dtype=np.float64).reshape(T) for t in thetas])
        noises = np.concatenate([np.asarray(t["noise"], dtype=np.float64).reshape(T) for t in thetas])
        if self.mean_fn is None:
            yres = self.y_train  # (T, N): entry b reads row b % T
        else:
            yres = np.concatenate([self._residual(t) for t in thetas])
        lml, info, grad, alpha = eng.fit_batch(self._kind, ells, scales, noises, jitter, yres, want_grad=True)
        out = []
        for c, u in enumerate(us):
            sl = slice(c * T, (c + 1) * T)
            if np.any(info[sl] != 0) or not np.all(np.isfinite(lml[sl])):
                out.append((-np.inf, np.zeros_like(u)))
            else:
                out.append(self._chain_rule(sites, u, thetas[c], float(np.sum(lml[sl])), self._glik(grad[sl]),
                                            alpha[sl], jacobian))
        return out

    # -- posterior for one sample of the parameters (vgp.py:122-176) ------------------------------------
    def get_mvn_posterior(self, X_new: np.ndarray, params: Dict[str, np.ndarray], noiseless: bool = False,
                          **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        """mean (T, M) and cov (T, M, M) of the T task posteriors for a single sample of GP parameters."""
        X_new = self._set_data(X_new)
        jitter = float(kwargs.get("jitter", 1e-6))
        T = self._tasks
        ells = self._ells(params)
        scales = np.broadcast_to(np.asarray(params["k_scale"], dtype=np.float64).reshape(-1), (T,))
        noises = np.broadcast_to(np.asarray(params["noise"], dtype=np.float64).reshape(-1), (T,))
        yres = self._residual(params)
        eng = _lib.get_engine(self._device)
        eng._train_owner = None  # the shared context is re-pointed at one task at a time below
        means, covs = [], []
        for t in range(T):
            eng.set_train(self.X_train[t])
            _, info = eng.factor(self._kind, ells[t], float(scales[t]), float(noises[t]), jitter, yres[t])
            noise_p = float(noises[t]) * (1 - int(bool(noiseless)))
            mean, cov, _ = eng.posterior(X_new[t], noise_p, jitter, want_cov=True)
            if info != 0:
                mean, cov = np.full_like(mean, np.nan), np.full_like(cov, np.nan)
            means.append(mean)
            covs.append(cov)
        mean, cov = np.stack(means), np.stack(covs)
        if self.mean_fn is not None:
            mean = mean + self._mean(X_new, params)
        return mean, cov

    # -- predict (gp.py:351-399 through vgp.py's task vmap) ---------------------------------------------
    def predict(self, rng_key, X_new: np.ndarray, samples: Optional[Dict[str, np.ndarray]] = None, n: int = 1,
                filter_nans: bool = False, noiseless: bool = False, device=None,
                **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        """Returns y_mean (T, M) and y_sampled (S, n, T, M)."""
        X_new = self._set_data(X_new)
        if samples is None:
            samples = self.get_samples(chain_dim=False)
        if isinstance(device, int):
            self._device = device
        jitter = float(kwargs.get("jitter", 1e-6))
        T, M = self._tasks, X_new.shape[1]
        S = len(next(iter(samples.values())))
        per = [{k: np.asarray(v)[s] for k, v in samples.items()} for s in range(S)]
        ells = np.concatenate([self._ells(p) for p in per])                        # entries ordered [s][t]
        scales = np.asarray(samples["k_scale"], dtype=np.float64).reshape(S * T)
        noises = np.asarray(samples["noise"], dtype=np.float64).reshape(S * T)
        mean_shift = None
        if self.mean_fn is not None:
            yres = np.concatenate([self._residual(p) for p in per])                # (S*T, N)
            mean_shift = np.stack([self._mean(X_new, p) for p in per])             # (S, T, M)
        else:
            yres = self.y_train                                                    # (T, N)
        eps = rng_from_key(rng_key).standard_normal((S, n, T, M))
        eng = self._engine()
        means, draws, infos = eng.predict_sweep(self._kind, ells, scales, noises, yres, X_new, noiseless, jitter,
                                                np.ascontiguousarray(eps.transpose(0, 2, 1, 3)).reshape(S * T, n, M))
        means = means.reshape(S, T, M)
        y_sampled = draws.reshape(S, T, n, M).transpose(0, 2, 1, 3)
        if mean_shift is not None:
            means = means + mean_shift
            y_sampled = y_sampled + mean_shift[:, None]
        if filter_nans:
            keep = ~np.isnan(y_sampled).any(axis=(1, 2, 3))
            y_sampled = y_sampled[keep]
        return means.mean(0), np.ascontiguousarray(y_sampled)

    def predict_in_batches(self, rng_key, X_new, batch_size=100, samples=None, n=1, filter_nans=False,
                           predict_fn=None, noiseless=False, device=None, **kwargs):
        """predict() over slices of the M axis of X_new (vgp.py:178-197)."""
        X_new = self._set_data(X_new)
        y_pred, y_sampled = self._predict_in_batches(rng_key, X_new, batch_size, 1, samples, n, filter_nans,
                                                     predict_fn, noiseless, device, **kwargs)
        return np.concatenate(y_pred, -1), np.concatenate(y_sampled, -1)

    def sample_from_prior(self, rng_key, X: np.ndarray, num_samples: int = 10):
        """Samples from the prior predictive distribution at X: (num_samples, T, N)."""
        X = self._set_data(X)
        rng = rng_from_key(rng_key)
        eng = _lib.get_engine(self._device)
        T, N = X.shape[0], X.shape[1]
        saved = self.X_train
        self.X_train = X  # the sites are sized by the task count of X
        try:
            out = np.empty((num_samples, T, N))
            for i in range(num_samples):
                theta = {s.name: (s.dist.sample(rng, s.shape) if s.shape else float(s.dist.sample(rng)))
                         for s in self._sites()}
                ells = self._ells(theta)
                loc = self._mean(X, theta) if self.mean_fn is not None else np.zeros((T, N))
                for t in range(T):
                    K = eng.gram(self._kind, X[t], X[t], ells[t], float(theta["k_scale"][t]),
                                 float(theta["noise"][t]) + 1e-6, True)
                    L, info = eng.potrf(K)
                    out[i, t] = loc[t] + L @ rng.standard_normal(N) if info == 0 else np.nan
        finally:
            self.X_train = saved
        return out
