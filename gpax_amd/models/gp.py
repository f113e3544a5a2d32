"""
ExactGP — fully Bayesian exact Gaussian process with the reference's method surface
(gpax/models/gp.py:30-432), computed by libgpx on an MI355X.

What changed underneath:
  * the NumPyro model + NUTS (gp.py:137-220) become an explicit log-joint whose likelihood term
    and analytic gradient come from gpx_factor / gpx_lml_grad, sampled by the host NUTS in
    gpax_amd/infer/nuts.py;
  * get_mvn_posterior (gp.py:253-277) uses POTRF + TRSM instead of the explicit inverse;
  * predict (gp.py:351-399) runs the vmap over posterior samples as the device-resident sweep
    gpx_predict_sweep (one theta in flight, nothing S*N*N is materialised);
  * priors are gpax_amd.infer.dist objects (NumPyro is not a dependency); `kernel_prior` /
    `noise_prior` (callables that run numpyro.sample) are not supported;
  * rng_key is an opaque seed (utils.rng_from_key); JAX threefry streams are not reproduced.
"""
from __future__ import annotations

import sys
import warnings
from typing import Callable, Dict, Optional, Tuple, Union

import numpy as np

from .. import _lib
from ..infer import dist
from ..infer.nuts import run_nuts
from ..kernels.kernels import kernel_name
from ..utils import threefry as _threefry
from ..utils.utils import rng_from_key, split_in_batches

kernel_fn_type = Callable[[np.ndarray, np.ndarray, Dict[str, np.ndarray], np.ndarray], np.ndarray]


class _Site:
    def __init__(self, name: str, shape: Tuple[int, ...], distribution: dist.Distribution):
        self.name, self.shape, self.dist = name, tuple(shape), distribution
        self.size = int(np.prod(shape)) if shape else 1


class _Lockstep:
    """Rendezvous of concurrently running chains: each chain thread submits the point where it needs
    log p and its gradient and blocks; when every live chain has submitted, the last one to arrive
    evaluates all requests as one batch and wakes the others.  Chains are independent, so results do
    not depend on the grouping."""

    def __init__(self, n: int, batch_fn):
        import threading
        self._cv = threading.Condition()
        self._live = n
        self._pending = {}
        self._results = {}
        self._batch_fn = batch_fn
        self._error = None

    def _run_batch(self):  # called with the lock held and every live chain waiting
        ids = sorted(self._pending)
        try:
            outs = self._batch_fn([self._pending[i] for i in ids])
            for i, o in zip(ids, outs):
                self._results[i] = o
        except Exception as ex:  # wake everybody up; each waiter re-raises
            self._error = ex
            for i in ids:
                self._results[i] = None
        self._pending.clear()
        self._cv.notify_all()

    def evaluate(self, cid: int, u):
        with self._cv:
            self._pending[cid] = u
            if len(self._pending) >= self._live:
                self._run_batch()
            else:
                while cid not in self._results:
                    self._cv.wait()
            out = self._results.pop(cid)
            if out is None:
                raise self._error
            return out

    def finish(self, cid: int):
        with self._cv:
            self._live -= 1
            if self._pending and len(self._pending) >= self._live:
                self._run_batch()


class ExactGP:

    """
    Gaussian process class

    Args:
        input_dim: number of input (feature) dimensions
        kernel: 'RBF' or 'Matern' (or gpax_amd.kernels.RBFKernel / MaternKernel)
        mean_fn: optional deterministic mean function  mean_fn(X) or mean_fn(X, params)
        kernel_prior: not supported on this path (NumPyro-program callable in the reference)
        mean_fn_prior: dict name -> distribution (or a callable returning one) for the mean-function
            parameters; the reference takes a callable that runs numpyro.sample
        noise_prior: not supported (deprecated in the reference)
        noise_prior_dist: prior on the noise variance (default LogNormal(0, 1))
        lengthscale_prior_dist: prior on the lengthscales (default LogNormal(0, 1))
    """

    _ride_along_bytes = 12e9  # device budget for the k_pX rows of one predict_in_batches sweep (per sample)

    def __init__(
        self,
        input_dim: int,
        kernel: Union[str, kernel_fn_type],
        mean_fn: Optional[Callable] = None,
        kernel_prior: Optional[Callable] = None,
        mean_fn_prior=None,
        noise_prior: Optional[Callable] = None,
        noise_prior_dist: Optional[dist.Distribution] = None,
        lengthscale_prior_dist: Optional[dist.Distribution] = None,
    ) -> None:
        if noise_prior is not None:
            warnings.warn("`noise_prior` is deprecated in gpax; use `noise_prior_dist`.", FutureWarning)
            raise NotImplementedError("`noise_prior` callables run numpyro.sample and have no MI355X path; "
                                      "pass `noise_prior_dist` (gpax_amd.dist.*) instead")
        if kernel_prior is not None:
            raise NotImplementedError("`kernel_prior` callables run numpyro.sample and have no MI355X path; "
                                      "pass `lengthscale_prior_dist` (gpax_amd.dist.*) instead")
        self.kernel_dim = input_dim
        self.kernel_name = kernel_name(kernel)
        self.kernel = kernel
        self._kind = _lib.kernel_kind(self.kernel_name)
        self.mean_fn = mean_fn
        self.kernel_prior = None
        self.mean_fn_prior = mean_fn_prior
        self.noise_prior = None
        self.noise_prior_dist = noise_prior_dist
        self.lengthscale_prior_dist = lengthscale_prior_dist
        self.X_train = None
        self.y_train = None
        self.mcmc = None
        self._samples = None
        self._chain_shape = None
        self._device = None
        self._data_version = 0

    # ------------------------------------------------------------------------------------------
    # model definition: sites, transforms, log-joint  (gp.py:137-164, 222-247)
    # ------------------------------------------------------------------------------------------
    def _mean_prior_dict(self) -> Dict[str, dist.Distribution]:
        if self.mean_fn_prior is None:
            return {}
        pri = self.mean_fn_prior() if callable(self.mean_fn_prior) else self.mean_fn_prior
        if not isinstance(pri, dict) or not all(isinstance(v, dist.Distribution) for v in pri.values()):
            raise NotImplementedError("mean_fn_prior must be a dict name -> gpax_amd.dist distribution "
                                      "(or a callable returning one)")
        return pri

    def model(self, X, y=None, **kwargs) -> None:
        """The reference's NumPyro program (gp.py:137-164).  There is no NumPyro here: the same log joint —
        priors of `_sites()` plus the MVN log-likelihood — is evaluated by `_log_joint` on the device."""
        raise NotImplementedError("ExactGP.model is a NumPyro program in the reference; gpax_amd evaluates the same "
                                  "log joint on the MI355X through _log_joint (host NUTS / SVI drive it)")

    def _sites(self):
        length_dist = self.lengthscale_prior_dist if self.lengthscale_prior_dist is not None else dist.LogNormal(0.0, 1.0)
        noise_dist = self.noise_prior_dist if self.noise_prior_dist is not None else dist.LogNormal(0.0, 1.0)
        sites = [_Site("k_length", (self.kernel_dim,), length_dist),  # plate "ard", gp.py:238-239
                 _Site("k_scale", (), dist.LogNormal(0.0, 1.0))]     # gp.py:241
        if self.kernel_name == "Periodic":
            sites.append(_Site("period", (), dist.LogNormal(0.0, 1.0)))  # gp.py:243-244
        sites.append(_Site("noise", (), noise_dist))                     # gp.py:222-227
        for name, d in self._mean_prior_dict().items():
            sites.append(_Site(name, (), d))
        return sites

    def _engine(self) -> _lib.Engine:
        """The shared context of this model's device with X_train resident.  Residency is keyed on (model, data
        version): `_data_version` is bumped by every fit() / _set_training_data(), so an X array mutated in place
        or a recycled buffer is uploaded again; Engine.set_train itself drops the previous owner."""
        eng = _lib.get_engine(self._device)
        key = (self._data_version, id(self.X_train))
        if getattr(eng, "_train_owner", None) is not self or getattr(eng, "_train_version", None) != key:
            eng.set_train(self.X_train)
            eng._train_owner = self
            eng._train_version = key
        return eng

    def _ell(self, params) -> np.ndarray:
        """d lengthscales (+ the period for the periodic kernel), as the C-ABI takes them."""
        return _lib.pack_ell(self._kind, params["k_length"], self.kernel_dim, params.get("period"))

    def _mean(self, X, params) -> np.ndarray:
        if self.mean_fn is None:
            return np.zeros(X.shape[0])
        args = [X, params] if self.mean_fn_prior is not None else [X]
        return np.asarray(self.mean_fn(*args), dtype=np.float64).squeeze()

    def _unpack(self, sites, u):
        theta, off = {}, 0
        for s in sites:
            ui = u[off:off + s.size]
            x = s.dist.transform(ui)
            theta[s.name] = x.reshape(s.shape) if s.shape else float(x[0])
            off += s.size
        return theta

    def _log_joint(self, sites, u, jitter: float, jacobian: bool, want_grad: bool = True, eng=None):
        """log p(y | theta) + log p(theta) [+ log |dtheta/du|] at theta = T(u) and its gradient
        w.r.t. u.  Returns (value, grad) — (-inf, zeros) when K(theta) is not positive definite.
        `eng`: a libgpx context that already holds X_train; default: the shared one."""
        if want_grad:  # value + gradient in ONE device call and one synchronisation (gpx_fit_batch with B = 1)
            return self._log_joint_batch(sites, [u], jitter, jacobian, eng=eng)[0]
        theta = self._unpack(sites, u)
        if eng is None:
            eng = self._engine()
        yres = self.y_train - self._mean(self.X_train, theta)
        lml, info = eng.factor(self._kind, self._ell(theta), theta["k_scale"], theta["noise"], jitter, yres)
        if info != 0 or not np.isfinite(lml):
            return -np.inf, np.zeros_like(u)
        return self._chain_rule(sites, u, theta, lml, None, None, jacobian)

    def _log_joint_batch(self, sites, us, jitter: float, jacobian: bool, eng=None):
        """_log_joint for a list of unconstrained vectors in ONE device pass (gpx_fit_batch: the chains of
        a multi-chain NUTS run advance in lockstep, the chain being a grid dimension of every launch)."""
        if eng is None:
            eng = self._engine()
        thetas = [self._unpack(sites, u) for u in us]
        ells = np.stack([self._ell(t) for t in thetas])
        scales = np.array([t["k_scale"] for t in thetas], dtype=np.float64)
        noises = np.array([t["noise"] for t in thetas], dtype=np.float64)
        if self.mean_fn is None:
            yres = self.y_train
        else:
            yres = np.stack([self.y_train - self._mean(self.X_train, t) for t in thetas])
        lml, info, grad, alpha = eng.fit_batch(self._kind, ells, scales, noises, jitter, yres, want_grad=True)
        out = []
        for b, u in enumerate(us):
            if info[b] != 0 or not np.isfinite(lml[b]):
                out.append((-np.inf, np.zeros_like(u)))
            else:
                out.append(self._chain_rule(sites, u, thetas[b], float(lml[b]), self._glik(grad[b]), alpha[b],
                                            jacobian))
        return out

    def _glik(self, g):
        """Device gradient rows [..., (d/d k_length.., (d/d period), d/d k_scale, d/d noise)] -> site name -> array."""
        if g is None:
            return None
        d_ = self.kernel_dim
        ne = d_ + (1 if self.kernel_name == "Periodic" else 0)
        glik = {"k_length": g[..., :d_], "k_scale": g[..., ne], "noise": g[..., ne + 1]}
        if self.kernel_name == "Periodic":
            glik["period"] = g[..., d_]
        return glik

    def _lognormal_plan(self, sites):
        """(loc, scale, log(scale) + log(2 pi)/2) per element of u when every site has a LogNormal prior, else None."""
        key = tuple((s.name, s.size, type(s.dist).__name__, float(getattr(s.dist, "loc", np.nan)),
                     float(getattr(s.dist, "scale", np.nan))) if type(s.dist) is dist.LogNormal
                    else (s.name, s.size, type(s.dist).__name__, id(s.dist)) for s in sites)
        cached = getattr(self, "_ln_plan", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        plan = None
        if sites and all(type(s.dist) is dist.LogNormal for s in sites):
            loc = np.concatenate([np.full(s.size, s.dist.loc) for s in sites])
            scale = np.concatenate([np.full(s.size, s.dist.scale) for s in sites])
            plan = (loc, scale, np.log(scale) + 0.5 * np.log(2 * np.pi))
        self._ln_plan = (key, plan)
        return plan

    def _chain_rule(self, sites, u, theta, lml, glik, alpha, jacobian: bool):
        """Add the log-priors (and log-Jacobians) to the device log-likelihood and map its gradient
        (glik: site name -> d lml / d site, any shape matching the site) to the unconstrained vector u."""
        plan = self._lognormal_plan(sites)
        if plan is not None and (glik is None or all(s.name in glik for s in sites)):
            # every site LogNormal (the default priors): the whole pass as a handful of vector operations, element
            # for element the arithmetic of the generic loop below (this runs once per leapfrog)
            loc, scale, const = plan
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                x = np.exp(u)
                lx = np.log(x)
                z = (lx - loc) / scale
                val = lml + float((-0.5 * z * z - const - lx).sum())
                if jacobian:
                    val += float(u.sum())
                if glik is None:
                    return val, np.zeros_like(u)
                gx = np.concatenate([np.asarray(glik[s.name], dtype=np.float64).reshape(-1) for s in sites])
                gu = (gx + (-(lx - loc) / scale ** 2 - 1.0) / x) * x
                if jacobian:
                    gu = gu + 1.0
            return val, gu
        val = lml
        grad = np.zeros_like(u)
        want_grad = glik is not None
        off = 0
        # one errstate for the whole pass (x = exp(u) can underflow far out in the tails: -inf / NaN terms make NUTS
        # treat the point as divergent); this runs once per leapfrog, so per-call overhead matters
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            for s in sites:
                ui = u[off:off + s.size]
                d_ = s.dist
                x = d_.transform(ui)
                val += float(d_.log_prob(x).sum())
                if jacobian:
                    lj, dlj = d_.log_abs_det_jacobian(ui)
                    val += float(lj.sum())
                if want_grad:
                    gl = glik.get(s.name)
                    if gl is not None:
                        gx = np.asarray(gl, dtype=np.float64).reshape(-1)
                    else:  # mean-function parameter: d lml / d phi = sum_i alpha_i d m_i / d phi
                        h = 1e-6 * max(1.0, abs(float(x[0])))
                        tp, tm = dict(theta), dict(theta)
                        tp[s.name] = float(x[0]) + h
                        tm[s.name] = float(x[0]) - h
                        dm = (self._mean(self.X_train, tp) - self._mean(self.X_train, tm)) / (2 * h)
                        gx = np.array([float(np.sum(alpha * dm))])
                    gu = (gx + d_.grad_log_prob(x)) * d_.dx_du(ui)
                    if jacobian:
                        gu = gu + dlj
                    grad[off:off + s.size] = gu
                off += s.size
        return val, grad

    def _init_unconstrained(self, sites, rng, num_samples: int = 10):
        """init_to_median(num_samples=10) (gp.py:207): per-site median of prior draws."""
        parts = []
        for s in sites:
            draws = s.dist.sample(rng, (num_samples, s.size))
            parts.append(s.dist.inverse(np.median(draws, axis=0)))
        return np.concatenate(parts)

    # ------------------------------------------------------------------------------------------
    # fit  (gp.py:166-220)
    # ------------------------------------------------------------------------------------------
    def fit(
        self,
        rng_key,
        X: np.ndarray,
        y: np.ndarray,
        num_warmup: int = 2000,
        num_samples: int = 2000,
        num_chains: int = 1,
        chain_method: str = "sequential",
        progress_bar: bool = True,
        print_summary: bool = True,
        device=None,
        **kwargs: float,
    ) -> None:
        """Run Hamiltonian Monte Carlo (NUTS) to infer the GP parameters.  `device` is a GPU
        ordinal (None: $LOCAL_RANK or 0).  **jitter: diagonal jitter (default 1e-6)."""
        X, y = self._set_data(X, y)
        self._device = device if isinstance(device, int) else None
        self.X_train = X
        self.y_train = y
        self._data_version += 1  # re-upload X even when the caller reuses (or mutated) the same array object
        self._ln_plan = None     # prior plan rebuilt once per fit
        jitter = float(kwargs.get("jitter", 1e-6))
        rng = rng_from_key(rng_key)
        sites = self._sites()
        # one independent generator per chain (children of the key): chains do not depend on how they are
        # scheduled.  chain_method 'parallel' / 'vectorized' advance the chains together, their gradient
        # requests batched into one device pass per round (the reference pmaps / vmaps them,
        # gp.py:173-174,214); 'sequential' runs them one after the other.
        chain_rngs = [rng] if num_chains == 1 else [np.random.default_rng(sd) for sd in rng.integers(0, 2 ** 63, num_chains)]
        concurrent = chain_method != "sequential" and num_chains > 1
        results = [None] * num_chains
        errors = []
        lockstep = _Lockstep(num_chains, lambda us: self._log_joint_batch(sites, us, jitter, jacobian=True)) \
            if concurrent else None

        def run_chain(c):
            try:
                crng = chain_rngs[c]

                def potential(u):
                    if lockstep is not None:
                        v, g = lockstep.evaluate(c, u)
                    else:
                        v, g = self._log_joint(sites, u, jitter, jacobian=True)
                    return (-v, -g) if np.isfinite(v) else (np.inf, np.zeros_like(u))

                u0 = None
                for _ in range(100):  # like NumPyro: redraw until the initial potential is finite
                    u0 = self._init_unconstrained(sites, crng)
                    if np.isfinite(potential(u0)[0]):
                        break
                prog = _Progress(progress_bar and (not concurrent or c == 0),
                                 f"chain {c + 1}/{num_chains}" if num_chains > 1 else "sample")
                results[c] = run_nuts(potential, u0, num_warmup, num_samples, crng, progress=prog)
                prog.close()
            except Exception as ex:
                errors.append(ex)
            finally:
                if lockstep is not None:
                    lockstep.finish(c)

        if concurrent:
            # the chains advance in lockstep: every round, each live chain asks for one gradient and the
            # requests are evaluated as ONE batched device pass (the chain is a grid dimension)
            import threading
            self._engine()  # X_train resident on the shared context before the chains start
            ts = [threading.Thread(target=run_chain, args=(c,)) for c in range(num_chains)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        else:
            for c in range(num_chains):
                run_chain(c)
        if errors:
            raise errors[0]
        chains = [r["draws"] for r in results]
        stats = [{k: v for k, v in r.items() if k != "draws"} for r in results]
        draws = np.stack(chains)  # (chains, S, dim)
        samples = {}
        off = 0
        for s in sites:
            ui = draws[:, :, off:off + s.size]
            x = s.dist.transform(ui)
            samples[s.name] = x.reshape(x.shape[:2] + tuple(s.shape)) if s.shape else x[..., 0]
            off += s.size
        self._samples = samples
        self._chain_shape = draws.shape[:2]
        self.mcmc = _MCMCResult(self, stats)
        if print_summary:
            self._print_summary()

    def get_samples(self, chain_dim: bool = False) -> Dict[str, np.ndarray]:
        """Posterior samples (gp.py:249-251): leading axis S, or (chains, S) when chain_dim."""
        if self._samples is None:
            raise RuntimeError("call fit() first")
        if chain_dim:
            return dict(self._samples)
        return {k: v.reshape((-1,) + v.shape[2:]) for k, v in self._samples.items()}

    # ------------------------------------------------------------------------------------------
    # posterior for one sample  (gp.py:253-293)
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _scalar(v) -> float:
        return float(np.asarray(v, dtype=np.float64).reshape(-1)[0])

    def get_mvn_posterior(self, X_new: np.ndarray, params: Dict[str, np.ndarray], noiseless: bool = False,
                          **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        """Mean and covariance of the multivariate normal posterior for a single sample of GP
        parameters (gp.py:253-277)."""
        X_new = self._set_data(X_new)
        jitter = float(kwargs.get("jitter", 1e-6))
        noise = self._scalar(params["noise"])
        noise_p = noise * (1 - int(bool(noiseless)))
        y_residual = self.y_train - self._mean(self.X_train, params)
        eng = self._engine()
        lml, info = eng.factor(self._kind, self._ell(params), self._scalar(params["k_scale"]), noise, jitter,
                               y_residual)
        mean, cov, _ = eng.posterior(X_new, noise_p, jitter, want_cov=True)
        if info != 0:
            mean = np.full_like(mean, np.nan)
            cov = np.full_like(cov, np.nan)
        if self.mean_fn is not None:
            mean = mean + self._mean(X_new, params)
        return mean, cov

    def _predict(self, rng_key, X_new: np.ndarray, params: Dict[str, np.ndarray], n: int, noiseless: bool = False,
                 **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        """Prediction with a single sample of GP parameters (gp.py:279-293)."""
        y_mean, K = self.get_mvn_posterior(X_new, params, noiseless, **kwargs)
        eps = rng_from_key(rng_key).standard_normal((n, y_mean.shape[0]))
        if not np.all(np.isfinite(K)):
            return y_mean, np.full((n, y_mean.shape[0]), np.nan)
        draws, info = self._engine().mvn_draw(eps)  # mean-function-free draw: loc + L eps
        if self.mean_fn is not None:
            draws = draws + self._mean(self._set_data(X_new), params)[None, :]
        return y_mean, draws

    # ------------------------------------------------------------------------------------------
    # predict  (gp.py:295-399)
    # ------------------------------------------------------------------------------------------
    def _predict_in_batches(self, rng_key, X_new, batch_size=100, batch_dim=0, samples=None, n=1, filter_nans=False,
                            predict_fn=None, noiseless=False, device=None, **kwargs):
        if predict_fn is None:
            predict_fn = lambda xi: self.predict(rng_key, xi, samples, n, filter_nans, noiseless, device, **kwargs)
        y_out1, y_out2 = [], []
        for Xi in split_in_batches(X_new, batch_size, dim=batch_dim):
            out1, out2 = predict_fn(Xi)
            y_out1.append(out1)
            y_out2.append(out2)
        return y_out1, y_out2

    def predict_in_batches(self, rng_key, X_new, batch_size=100, samples=None, n=1, filter_nans=False,
                           predict_fn=None, noiseless=False, device=None, **kwargs):
        """predict() over slices of X_new (gp.py:325-349); same rng_key for every slice.
        With the default predict the slices become covariance blocks of ONE sweep: K(theta) is factored once per
        sample instead of once per sample and slice (identical values: every slice's posterior and draws are what
        predict() on that slice alone returns)."""
        if predict_fn is None and type(self).predict is ExactGP.predict:
            X_new = self._set_data(X_new)
            bs = max(1, int(batch_size))
            # all slices of one call ride along in the factorisation: (N + M) x N doubles per sample on the device.
            # Very large grids go in groups of slices sized to ~12 GB of ride-along rows (same values either way).
            rows_max = max(bs, int(self._ride_along_bytes / (8.0 * (self.X_train.shape[0] + 256))) // bs * bs)
            if X_new.shape[0] <= rows_max:
                return self.predict(rng_key, X_new, samples, n, filter_nans, noiseless, device, _m_slice=bs, **kwargs)
            outs = [self.predict(rng_key, X_new[i:i + rows_max], samples, n, False, noiseless, device, _m_slice=bs,
                                 **kwargs) for i in range(0, X_new.shape[0], rows_max)]
            y_pred = np.concatenate([o[0] for o in outs], 0)
            y_sampled = np.concatenate([o[1] for o in outs], -1)
            if filter_nans:
                y_sampled = y_sampled[~np.isnan(y_sampled).any(axis=(1, 2))]
            return y_pred, y_sampled
        y_pred, y_sampled = self._predict_in_batches(rng_key, X_new, batch_size, 0, samples, n, filter_nans,
                                                     predict_fn, noiseless, device, **kwargs)
        return np.concatenate(y_pred, 0), np.concatenate(y_sampled, -1)

    def predict(self, rng_key, X_new: np.ndarray, samples: Optional[Dict[str, np.ndarray]] = None, n: int = 1,
                filter_nans: bool = False, noiseless: bool = False, device=None,
                **kwargs: float) -> Tuple[np.ndarray, np.ndarray]:
        """
        Make prediction at X_new points using posterior samples for GP parameters (gp.py:351-399).

        Returns the centre of mass of the sampled means (M,) and all sampled predictions (S, n, M).
        """
        X_new = self._set_data(X_new)
        if samples is None:
            samples = self.get_samples(chain_dim=False)
        if isinstance(device, int):
            self._device = device
        jitter = float(kwargs.get("jitter", 1e-6))
        m_slice = int(kwargs.pop("_m_slice", 0))  # predict_in_batches: covariance blocks of this many test points
        ells, scales, noises, yres, eps, mean_shift = self._sweep_inputs(rng_key, X_new, samples, n, m_slice)
        # several samples in flight per GPU: independent libgpx contexts on the same device
        engines = _lib.get_sweep_engines(self._device)  # concurrent_sweep re-uploads X (set_train drops ownership)
        means, y_sampled, infos = _lib.concurrent_sweep(engines, self.X_train, self._kind, ells, scales, noises, yres,
                                                        X_new, noiseless, jitter, eps, m_slice=m_slice)
        return self._sweep_outputs(means, y_sampled, mean_shift, filter_nans)

    def _sweep_inputs(self, rng_key, X_new, samples, n, m_slice: int = 0):
        """Per-sample tables of the predictive sweep: packed lengthscales (S, n_ell), scales, noises, the
        residual(s) y - m(X), the standard normals of the draws and the mean-function shift at X_new."""
        S = len(next(iter(samples.values())))
        d, M = self.kernel_dim, X_new.shape[0]
        ells = np.asarray(samples["k_length"], dtype=np.float64).reshape(S, -1)
        ells = np.ascontiguousarray(np.broadcast_to(ells, (S, d)))
        if self.kernel_name == "Periodic":
            ells = np.ascontiguousarray(np.concatenate(
                [ells, np.asarray(samples["period"], dtype=np.float64).reshape(S, 1)], axis=1))
        scales = np.asarray(samples["k_scale"], dtype=np.float64).reshape(S)
        noises = np.asarray(samples["noise"], dtype=np.float64).reshape(S)
        mean_shift = None
        if self.mean_fn is not None:
            per = [{k: np.asarray(v)[s] for k, v in samples.items()} for s in range(S)]
            yres = np.stack([self.y_train - self._mean(self.X_train, p) for p in per])
            mean_shift = np.stack([self._mean(X_new, p) for p in per])
        else:
            yres = self.y_train
        # a threefry key reproduces the reference's stream: split over the S samples, normal(key_s, (n, M)) each
        draw = (lambda m: _threefry.predict_normals(rng_key, S, n, m)) if isinstance(rng_key, _threefry.ThreefryKey) \
            else (lambda m: rng_from_key(rng_key).standard_normal((S, n, m)))
        if 0 < m_slice < M:  # the same key for every slice, as predict_in_batches passes it (gp.py:344-347)
            eps = np.concatenate([draw(min(m_slice, M - m0)) for m0 in range(0, M, m_slice)], axis=-1)
        else:
            eps = draw(M)
        return ells, scales, noises, yres, eps, mean_shift

    @staticmethod
    def _sweep_outputs(means, y_sampled, mean_shift, filter_nans):
        if mean_shift is not None:
            means = means + mean_shift
            y_sampled = y_sampled + mean_shift[:, None, :]
        if filter_nans:
            keep = ~np.isnan(y_sampled).any(axis=(1, 2))
            y_sampled = y_sampled[keep]
        return means.mean(0), y_sampled

    def predict_distributed(self, rng_key, X_new: np.ndarray, samples: Optional[Dict[str, np.ndarray]] = None,
                            n: int = 1, filter_nans: bool = False, noiseless: bool = False,
                            **kwargs: float) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        """predict() with the posterior samples sharded over the ranks of an initialised torch.distributed
        group (one process per GPU, `python -m torch.distributed.run`; backend nccl = RCCL over xGMI).
        A collective call: every rank enters it with a model of the same configuration; only rank 0 needs the
        fitted state / arguments — they are broadcast, each rank sweeps its contiguous block of samples on its
        own GPU, and rank 0 returns (y_mean, y_sampled); the other ranks return None (gp.py:351-399's vmap axis
        is the only axis of the path that shards)."""
        from ..parallel import Communicator, predict_sharded
        comm = Communicator()
        jitter = float(kwargs.get("jitter", 1e-6))
        mean_shift = None
        if comm.rank == 0:
            X_new = self._set_data(X_new)
            if samples is None:
                samples = self.get_samples(chain_dim=False)
            ells, scales, noises, yres, eps, mean_shift = self._sweep_inputs(rng_key, X_new, samples, n)
            args = (self.X_train, yres, X_new, {"k_length": ells, "k_scale": scales, "noise": noises}, eps)
        else:
            args = (None, None, None, None, None)
        engines = _lib.get_sweep_engines(self._device)
        res = predict_sharded(engines, self._kind, *args, noiseless, jitter, comm)
        if res is None:
            return None
        means, y_sampled, infos = res
        return self._sweep_outputs(means, y_sampled, mean_shift, filter_nans)

    def sample_from_prior(self, rng_key, X: np.ndarray, num_samples: int = 10):
        """Samples from the prior predictive distribution at X (gp.py:401-408)."""
        X = self._set_data(X)
        rng = rng_from_key(rng_key)
        eng = _lib.get_engine(self._device)
        out = np.empty((num_samples, X.shape[0]))
        for i in range(num_samples):
            theta = {s.name: (s.dist.sample(rng, s.shape) if s.shape else float(s.dist.sample(rng))) for s in self._sites()}
            K = eng.gram(self._kind, X, X, self._ell(theta), theta["k_scale"], theta["noise"] + 1e-6, True)
            L, info = eng.potrf(K)
            out[i] = self._mean(X, theta) + L @ rng.standard_normal(X.shape[0]) if info == 0 else np.nan
        return out

    # ------------------------------------------------------------------------------------------
    # data plumbing  (gp.py:410-432)
    # ------------------------------------------------------------------------------------------
    def _set_data(self, X, y=None):
        X = np.asarray(X, dtype=np.float64)
        X = X if X.ndim > 1 else X[:, None]
        X = np.ascontiguousarray(X)
        if y is not None:
            return X, np.ascontiguousarray(np.asarray(y, dtype=np.float64).squeeze())
        return X

    def _set_training_data(self, X_train_new=None, y_train_new=None, device=None) -> None:
        if X_train_new is not None:
            self.X_train = self._set_data(X_train_new)
        if y_train_new is not None:
            self.y_train = np.ascontiguousarray(np.asarray(y_train_new, dtype=np.float64).squeeze())
        if isinstance(device, int):
            self._device = device
        self._data_version += 1

    def _print_summary(self):
        samples = self.get_samples(chain_dim=True)
        print_summary(samples)


class _MCMCResult:
    """Stand-in for `self.mcmc` (the reference stores the numpyro MCMC object)."""

    def __init__(self, model, stats):
        self._model, self.stats = model, stats

    def get_samples(self, group_by_chain: bool = False):
        return self._model.get_samples(chain_dim=group_by_chain)

    def get_extra_fields(self):
        return self.stats


class _Progress:
    def __init__(self, enabled: bool, label: str):
        self.enabled, self.label, self.last = enabled, label, -1

    def __call__(self, it, total, info):
        if not self.enabled:
            return
        pct = (it + 1) * 100 // total
        if pct != self.last and (pct % 5 == 0 or it + 1 == total):
            self.last = pct
            extra = ", ".join(f"{k}={v:.3g}" if isinstance(v, float) else f"{k}={v}" for k, v in info.items())
            sys.stdout.write(f"\r{self.label}: {pct:3d}% [{it + 1}/{total}] {extra}   ")
            sys.stdout.flush()

    def close(self):
        if self.enabled:
            sys.stdout.write("\n")


def print_summary(samples: Dict[str, np.ndarray]) -> None:
    """Plain-text posterior summary (the reference prints numpyro.diagnostics.print_summary)."""
    print(f"\n{'':>16s}{'mean':>10s}{'std':>10s}{'median':>10s}{'5.0%':>10s}{'95.0%':>10s}")
    for name, v in samples.items():
        flat = np.asarray(v).reshape((-1,) + np.asarray(v).shape[2:])
        cols = flat.reshape(flat.shape[0], -1)
        for j in range(cols.shape[1]):
            c = cols[:, j]
            label = name if cols.shape[1] == 1 else f"{name}[{j}]"
            print(f"{label:>16s}{c.mean():10.2f}{c.std():10.2f}{np.median(c):10.2f}"
                  f"{np.quantile(c, 0.05):10.2f}{np.quantile(c, 0.95):10.2f}")
    print()
