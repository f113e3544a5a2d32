"""
VarNoiseGP — heteroskedastic GP with the reference's surface (gpax/models/hskgp.py:24-220): a latent GP over
the log noise variance at the training inputs ("log_var", an N-vector with an MVN prior under the noise
kernel) and the main GP whose training covariance is kernel + jitter I + diag(exp(log_var)).

One NUTS leapfrog = two device fit steps on the same context: the main likelihood with the per-point diagonal
exp(log_var) (gpx_set_diag; its derivative w.r.t. the diagonal is gpx_lml_grad_diag) and the MVN log-density
of log_var under the noise kernel (an exact-GP likelihood with y = log_var and noise 0; its derivative w.r.t.
log_var is -alpha).  predict = two batched sweeps: the noise GP's posterior mean at X_new per HMC sample
(-> predicted noise variance), then the main GP's sweep with that variance added to diag(cov)
(`pred_diag`, hskgp.py:188-204).  As in the reference, the main GP's predictive posterior conditions on
kernel + jitter I only — the inferred noise is NOT in its training block (hskgp.py:176-183); mirrored as is.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import numpy as np

from .. import _lib
from ..infer import dist
from ..utils.utils import rng_from_key
from .gp import ExactGP, _Site


class _Flat(dist.Distribution):
    """Placeholder 'prior' of the log_var site: its density is the MVN term added by the model itself."""
    support = "real"

    def log_prob(self, x):
        return np.zeros_like(np.asarray(x, dtype=np.float64))

    def grad_log_prob(self, x):
        return np.zeros_like(np.asarray(x, dtype=np.float64))

    def sample(self, rng, shape=()):
        return np.zeros(shape)  # prior median of log_var without a noise mean function

    def median(self):
        return 0.0


class VarNoiseGP(ExactGP):
    """
    Heteroskedastic Gaussian process class

    Args:
        input_dim, kernel, mean_fn, kernel_prior, mean_fn_prior, lengthscale_prior_dist: as ExactGP
        noise_kernel: kernel of the latent log-variance GP ('RBF', 'Matern')
        noise_mean_fn: optional (positive) noise mean function; its log is the prior mean of log_var
        noise_mean_fn_prior: dict name -> gpax_amd.dist distribution for its parameters
        noise_lengthscale_prior_dist: prior of k_noise_length (default LogNormal(0, 1))
    """

    _chains_over_devices = False  # two device passes per leapfrog on the shared context (set_diag state): one GPU


    def __init__(self, input_dim: int, kernel: str, noise_kernel: str = 'RBF', mean_fn: Optional[Callable] = None,
                 kernel_prior=None, mean_fn_prior=None, noise_kernel_prior=None,
                 lengthscale_prior_dist: Optional[dist.Distribution] = None, noise_mean_fn: Optional[Callable] = None,
                 noise_mean_fn_prior=None, noise_lengthscale_prior_dist: Optional[dist.Distribution] = None) -> None:
        super().__init__(input_dim, kernel, mean_fn, kernel_prior, mean_fn_prior, None, None, lengthscale_prior_dist)
        if noise_kernel_prior is not None:
            raise NotImplementedError("`noise_kernel_prior` callables run numpyro.sample and have no MI355X path")
        if noise_kernel not in ("RBF", "Matern"):
            raise NotImplementedError("noise_kernel must be 'RBF' or 'Matern' on the MI355X path")
        self.noise_kernel_name = noise_kernel
        self._noise_kind = _lib.kernel_kind(noise_kernel)
        self.noise_mean_fn = noise_mean_fn
        if noise_mean_fn_prior is not None and not isinstance(noise_mean_fn_prior, dict):
            raise NotImplementedError("noise_mean_fn_prior must be a dict name -> gpax_amd.dist distribution")
        self.noise_mean_fn_prior = noise_mean_fn_prior
        self.noise_lengthscale_prior_dist = noise_lengthscale_prior_dist

    # -- sites (hskgp.py:102-165) -----------------------------------------------------------------------
    def _sites(self):
        N = self.X_train.shape[0]
        # the noise kernel's default priors are the reference's program (hskgp.py:155-162), traced — overridable
        sites, _, _ = self._traced(self._sample_noise_kernel_params, "_sample_noise_kernel_params")
        if [(sx.name, sx.size) for sx in sites] != [("k_noise_scale", 1), ("k_noise_length", 1)]:
            raise ValueError("_sample_noise_kernel_params must sample the scalar sites k_noise_scale and k_noise_length")
        for name, d in (self.noise_mean_fn_prior or {}).items():
            sites.append(_Site(name, (), d))
        sites.append(_Site("log_var", (N,), _Flat()))                    # hskgp.py:131-134
        sites += [s for s in super()._sites() if s.name != "noise"]      # no inferred scalar noise
        return sites

    def model(self, X, y=None, params: Optional[Dict[str, np.ndarray]] = None, **kwargs: float) -> float:
        """What the reference's NumPyro program defines (hskgp.py:105-153), evaluated as ExactGP.model evaluates the exact
        one: site log-densities + the latent site's own density log N(log_var | log noise_mean_fn, k_noise) +
        log N(y | m, k + jitter I + diag(exp(log_var))) at `params` (default: prior medians, log_var = 0).  y = None:
        without the last term."""
        X = self._set_data(X)
        jitter = float(kwargs.get("jitter", 1e-6))
        yy = np.zeros(X.shape[0]) if y is None else np.asarray(y, dtype=np.float64).squeeze()
        with self._TrainingData(self, X, yy):
            sites = self._sites()
            theta, val = self._theta_and_log_prior(sites, params)
            if y is not None:
                v, _ = self._log_joint(sites, self._unconstrained(sites, theta), jitter, jacobian=False, want_grad=False)
                return float(v) if np.isfinite(v) else float("nan")
            eng = self._engine()
            eng.set_diag(None)
            lv_res = np.asarray(theta["log_var"], dtype=np.float64).reshape(-1) - self._noise_loc(X, theta)
            lml2, info2 = eng.factor(self._noise_kind, self._noise_ell(theta), self._scalar(theta["k_noise_scale"]), 0.0,
                                     jitter, lv_res)
            return val + lml2 if info2 == 0 else float("nan")

    def _sample_noise_kernel_params(self) -> Dict[str, np.ndarray]:
        """The reference's default priors of the noise kernel as a program (hskgp.py:155-162): k_noise_scale
        LogNormal(0, 1), k_noise_length noise_lengthscale_prior_dist or LogNormal(0, 1) — a scalar, no ARD plate."""
        from ..infer.primitives import sample
        nl = self.noise_lengthscale_prior_dist if self.noise_lengthscale_prior_dist is not None else dist.LogNormal(0.0, 1.0)
        noise_scale = sample("k_noise_scale", dist.LogNormal(0.0, 1.0))
        noise_length = sample("k_noise_length", nl)
        return {"k_noise_length": noise_length, "k_noise_scale": noise_scale}

    def _noise_loc(self, X, params) -> np.ndarray:
        """Prior mean of log_var: log(noise_mean_fn(X[, params])) or zeros (hskgp.py:120-128)."""
        if self.noise_mean_fn is None:
            return np.zeros(X.shape[0])
        args = [X, params] if self.noise_mean_fn_prior is not None else [X]
        return np.log(np.asarray(self.noise_mean_fn(*args), dtype=np.float64)).squeeze()

    def _noise_ell(self, params) -> np.ndarray:
        return np.full(self.kernel_dim, self._scalar(params["k_noise_length"]))

    # -- log joint and gradient -------------------------------------------------------------------------
    def _log_joint(self, sites, u, jitter: float, jacobian: bool, want_grad: bool = True, eng=None):
        theta = self._unpack(sites, u)
        if eng is None:
            eng = self._engine()
        bad = (-np.inf, np.zeros_like(u))
        lv = np.asarray(theta["log_var"], dtype=np.float64)
        with np.errstate(over="ignore"):
            v = np.exp(lv)
        if not np.all(np.isfinite(v)):
            return bad
        # main GP: y | theta, v   (hskgp.py:146-153)
        yres = self.y_train - self._mean(self.X_train, theta)
        eng.set_diag(v)
        try:  # the per-point diagonal must never outlive this block on the shared context, whatever happens inside
            lml1, info1 = eng.factor(self._kind, self._ell(theta), theta["k_scale"], 0.0, jitter, yres)
            if info1 != 0 or not np.isfinite(lml1):
                return bad
            g1 = a1 = gdiag = None
            if want_grad:
                g_ell, g_scale, _, a1 = eng.lml_grad()
                gdiag = eng.lml_grad_diag()
                g1 = np.concatenate([g_ell, [g_scale, 0.0]])
        finally:
            eng.set_diag(None)
        # noise GP: log_var | theta_n ~ MVN(noise_loc, k_noise)   (hskgp.py:129-134)
        lv_res = lv - self._noise_loc(self.X_train, theta)
        lml2, info2 = eng.factor(self._noise_kind, self._noise_ell(theta), self._scalar(theta["k_noise_scale"]), 0.0,
                                 jitter, lv_res)
        if info2 != 0 or not np.isfinite(lml2):
            return bad
        glik = None
        if want_grad:
            n_ell, n_scale, _, a2 = eng.lml_grad()
            glik = self._glik(g1)
            glik.pop("noise", None)
            glik["k_noise_length"] = np.array([np.sum(n_ell)])  # one lengthscale shared by all input dims
            glik["k_noise_scale"] = np.array([n_scale])
            glik["log_var"] = gdiag * v - a2
            for name in (self.noise_mean_fn_prior or {}):  # d lml2 / d phi = alpha2 . d loc / d phi
                x = float(theta[name])
                h = 1e-6 * max(1.0, abs(x))
                tp, tm = dict(theta), dict(theta)
                tp[name], tm[name] = x + h, x - h
                dloc = (self._noise_loc(self.X_train, tp) - self._noise_loc(self.X_train, tm)) / (2 * h)
                glik[name] = np.array([float(a2 @ dloc)])
        return self._chain_rule(sites, u, theta, lml1 + lml2, glik, a1, jacobian)

    def _log_joint_batch(self, sites, us, jitter: float, jacobian: bool, eng=None):
        # two dependent factorisations with a different per-point diagonal per chain: evaluated chain by chain
        return [self._log_joint(sites, u, jitter, jacobian, eng=eng) for u in us]

    # -- posterior for one sample (hskgp.py:167-206) ----------------------------------------------------
    def get_mvn_posterior(self, X_new: np.ndarray, params: Dict[str, np.ndarray], *args, **kwargs
                          ) -> Tuple[np.ndarray, np.ndarray]:
        """Main GP's predictive mean and the combined (main + predicted noise variance) covariance."""
        X_new = self._set_data(X_new)
        jitter = float(kwargs.get("jitter", 1e-6))
        eng = self._engine()
        eng.set_diag(None)
        yres = self.y_train - self._mean(self.X_train, params)
        _, info = eng.factor(self._kind, self._ell(params), self._scalar(params["k_scale"]), 0.0, jitter, yres)
        mean, cov, _ = eng.posterior(X_new, 0.0, jitter, want_cov=True)
        lv_res = np.asarray(params["log_var"], dtype=np.float64).reshape(-1) - self._noise_loc(self.X_train, params)
        _, info2 = eng.factor(self._noise_kind, self._noise_ell(params), self._scalar(params["k_noise_scale"]), 0.0,
                              jitter, lv_res)
        plv, _, _ = eng.posterior(X_new, 0.0, jitter, want_cov=False)
        plv = plv + self._noise_loc(X_new, params)
        if info != 0 or info2 != 0:
            return np.full_like(mean, np.nan), np.full_like(cov, np.nan)
        if self.mean_fn is not None:
            mean = mean + self._mean(X_new, params)
        return mean, cov + np.diag(np.exp(plv))

    def _predict(self, rng_key, X_new: np.ndarray, params: Dict[str, np.ndarray], n: int, noiseless: bool = False,
                 **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        y_mean, K = self.get_mvn_posterior(X_new, params, noiseless, **kwargs)
        eps = rng_from_key(rng_key).standard_normal((n, y_mean.shape[0]))
        if not np.all(np.isfinite(K)):
            return y_mean, np.full((n, y_mean.shape[0]), np.nan)
        L, info = self._engine().potrf(K)
        if info != 0:
            return y_mean, np.full((n, y_mean.shape[0]), np.nan)
        return y_mean, y_mean[None, :] + eps @ L.T

    # -- predict: two batched sweeps --------------------------------------------------------------------
    def predict(self, rng_key, X_new: np.ndarray, samples: Optional[Dict[str, np.ndarray]] = None, n: int = 1,
                filter_nans: bool = False, noiseless: bool = False, device=None,
                **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        X_new = self._set_data(X_new)
        if samples is None:
            samples = self.get_samples(chain_dim=False)
        if isinstance(device, int):
            self._device = device
        jitter = float(kwargs.get("jitter", 1e-6))
        S = len(next(iter(samples.values())))
        d, M = self.kernel_dim, X_new.shape[0]
        per = [{k: np.asarray(v)[s] for k, v in samples.items()} for s in range(S)]
        eng = self._engine()
        eng.set_diag(None)
        # 1) noise GP: predicted log variance at X_new for every sample
        n_ells = np.repeat(np.asarray(samples["k_noise_length"], dtype=np.float64).reshape(S, 1), d, axis=1)
        n_scales = np.asarray(samples["k_noise_scale"], dtype=np.float64).reshape(S)
        lv = np.asarray(samples["log_var"], dtype=np.float64).reshape(S, -1)
        if self.noise_mean_fn is not None:
            loc_X = np.stack([self._noise_loc(self.X_train, p) for p in per])
            loc_new = np.stack([self._noise_loc(X_new, p) for p in per])
        else:
            loc_X, loc_new = 0.0, 0.0
        plv, _, info_n = eng.predict_sweep(self._noise_kind, n_ells, n_scales, np.zeros(S), lv - loc_X, X_new, True, jitter,
                                           None)
        pred_var = np.exp(plv + loc_new)
        pred_var[info_n != 0] = np.nan
        # 2) main GP with the predicted variance on the diagonal of every sample's covariance
        ells = np.ascontiguousarray(np.broadcast_to(np.asarray(samples["k_length"], dtype=np.float64).reshape(S, -1), (S, d)))
        if self.kernel_name == "Periodic":
            ells = np.ascontiguousarray(np.concatenate(
                [ells, np.asarray(samples["period"], dtype=np.float64).reshape(S, 1)], axis=1))
        scales = np.asarray(samples["k_scale"], dtype=np.float64).reshape(S)
        mean_shift = None
        if self.mean_fn is not None:
            yres = np.stack([self.y_train - self._mean(self.X_train, p) for p in per])
            mean_shift = np.stack([self._mean(X_new, p) for p in per])
        else:
            yres = self.y_train
        eps = rng_from_key(rng_key).standard_normal((S, n, M))
        means, y_sampled, infos = eng.predict_sweep(self._kind, ells, scales, np.zeros(S), yres, X_new, True, jitter, eps,
                                                    pred_diag=np.nan_to_num(pred_var, nan=0.0))
        nanrow = np.isnan(pred_var).any(axis=1)
        means[nanrow] = np.nan
        y_sampled[nanrow] = np.nan
        if mean_shift is not None:
            means = means + mean_shift
            y_sampled = y_sampled + mean_shift[:, None, :]
        if filter_nans:
            keep = ~np.isnan(y_sampled).any(axis=(1, 2))
            y_sampled = y_sampled[keep]
        return means.mean(0), y_sampled

    def get_data_var_samples(self) -> np.ndarray:
        """Samples of the inferred (training) data variance — aka noise (hskgp.py:208-217)."""
        samples = self.get_samples()
        log_var = np.asarray(samples["log_var"], dtype=np.float64)
        if self.noise_mean_fn is not None:
            S = log_var.shape[0]
            per = [{k: np.asarray(v)[s] for k, v in samples.items()} for s in range(S)]
            log_var = log_var + np.stack([self._noise_loc(self.X_train, p) for p in per])
        return np.exp(log_var)

    def _print_summary(self):
        from .gp import print_summary
        samples = self.get_samples(chain_dim=True)
        print_summary({k: v for k, v in samples.items() if 'log_var' not in k})
