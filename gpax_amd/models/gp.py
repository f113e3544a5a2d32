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
  * priors are gpax_amd.infer.dist objects (NumPyro is not a dependency); `kernel_prior` / `noise_prior` /
    `mean_fn_prior` callables are written with gpax_amd.sample / plate instead of numpyro.sample / plate and are
    traced once for their sites (infer/primitives.py) — or given declaratively as a dict name -> distribution;
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
from ..infer.primitives import deterministic, plate, sample, trace_sites
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
        kernel_prior: custom priors over the kernel parameters: a function that calls gpax_amd.sample(name, dist)
            (inside gpax_amd.plate for vector sites) and returns {"k_length": ..., "k_scale": ...} — the reference's
            numpyro.sample callables with the import swapped — or a dict name -> distribution
        mean_fn_prior: the same for the parameters of mean_fn
        noise_prior: the same for the noise variance (deprecated in the reference in favour of noise_prior_dist)
        noise_prior_dist: prior on the noise variance (default LogNormal(0, 1))
        lengthscale_prior_dist: prior on the lengthscales (default LogNormal(0, 1))
        mean_fn_grad: optional mean_fn_grad(X, params) -> {name: d mean / d name (N,)}; without it the derivative
            of the mean function w.r.t. its parameters is taken by complex-step differentiation (exact to rounding
            for the analytic NumPy expressions mean functions are made of), falling back to central differences
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
        mean_fn_grad: Optional[Callable] = None,
    ) -> None:
        if noise_prior is not None:  # gp.py:108-115
            warnings.warn("`noise_prior` is deprecated and will be removed in a future version. Please use "
                          "`noise_prior_dist` instead, which accepts a gpax_amd.dist distribution, e.g. "
                          "`dist.HalfNormal(scale=0.1)`, rather than a function that calls `sample`.", FutureWarning)
        if kernel_prior is not None:  # gp.py:116-123
            warnings.warn("`kernel_prior` will remain available for complex priors. However, for modifying only the "
                          "lengthscales, it is recommended to use `lengthscale_prior_dist` instead.", UserWarning)
        self.kernel_dim = input_dim
        self.kernel_name = kernel_name(kernel)
        self.kernel = kernel
        self._kind = _lib.kernel_kind(self.kernel_name)
        self.mean_fn = mean_fn
        self.kernel_prior = kernel_prior
        self.mean_fn_prior = mean_fn_prior
        self.noise_prior = noise_prior
        self.mean_fn_grad = mean_fn_grad
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
    def _traced(self, fn, what):
        """Sites of a prior given as callable (gpax_amd.sample program) or dict name -> distribution."""
        sites, returned, det = trace_sites(fn, what)
        return [_Site(n, shp, dd) for n, shp, dd in sites], returned, det

    def _mean_prior_sites(self):
        if self.mean_fn_prior is None:
            return []
        return self._traced(self.mean_fn_prior, "mean_fn_prior")[0]

    def _mean_prior_dict(self) -> Dict[str, dist.Distribution]:
        return {s.name: s.dist for s in self._mean_prior_sites()}

    def model(self, X, y=None, params: Optional[Dict[str, np.ndarray]] = None, **kwargs) -> float:
        """The reference's NumPyro program (gp.py:137-164) registers the prior sites and the MVN likelihood with a
        tracer; there is no tracer here, so `model` evaluates what that program defines: the log joint
            log p(y | theta) + sum_sites log p(theta_site)
        at `params` (constrained values, site name -> value; default: the prior medians), the likelihood on the
        device (gpx_factor).  y = None: the log prior alone.  NaN when K(theta) is not positive definite."""
        X = self._set_data(X)
        sites = self._sites()
        theta, val = self._theta_and_log_prior(sites, params)
        if y is None:
            return val
        y = np.asarray(y, dtype=np.float64).squeeze()
        jitter = float(kwargs.get("jitter", 1e-6))
        eng = _lib.get_engine(self._device)
        eng.set_train(X)
        lml, info = eng.factor(self._kind, self._ell(theta), self._scalar(theta["k_scale"]),
                               self._scalar(theta["noise"]), jitter, y - self._mean(X, theta))
        return val + lml if info == 0 else float("nan")

    def _theta_and_log_prior(self, sites, params):
        """theta: site name -> constrained value (`params` where given, the prior median elsewhere, plus the deterministic
        sites) and the sum of the site log-densities there — the prior part of what `model` returns."""
        theta = {s.name: (np.full(s.shape, float(s.dist.median())) if s.shape else float(s.dist.median()))
                 for s in sites}
        if params is not None:
            theta.update({k: v for k, v in params.items() if k in theta})
        theta = self._with_deterministic(theta)
        val = 0.0
        with np.errstate(divide="ignore", invalid="ignore"):
            for s_ in sites:
                val += float(np.sum(s_.dist.log_prob(np.asarray(theta[s_.name], dtype=np.float64).reshape(-1))))
        return theta, val

    def _unconstrained(self, sites, theta) -> np.ndarray:
        """u with theta = T(u), site by site (the inverse of _unpack)."""
        return np.concatenate([np.asarray(s.dist.inverse(np.asarray(theta[s.name], dtype=np.float64).reshape(-1)),
                                          dtype=np.float64).reshape(-1) for s in sites])

    class _TrainingData:
        """`with model._TrainingData(model, X, y):` — the model's training data swapped for the block (what the `model`
        methods of the subclasses evaluate their log joint on), put back afterwards whatever happens inside."""

        def __init__(self, m, X, y):
            self.m, self.new = m, (X, y)

        def __enter__(self):
            m = self.m
            self.old = (m.X_train, m.y_train)
            m.X_train, m.y_train = self.new
            m._data_version += 1
            return m

        def __exit__(self, *exc):
            m = self.m
            m.X_train, m.y_train = self.old
            m._data_version += 1
            return False

    def _kernel_sites(self):
        """k_length (plate 'ard'), k_scale, period — default priors (gp.py:229-247) or the traced kernel_prior."""
        self._det = {}
        if self.kernel_prior is not None:
            sites, returned, det = self._traced(self.kernel_prior, "kernel_prior")
            have = set(returned) if isinstance(returned, dict) else set()
            if not {"k_length", "k_scale"} <= have:
                raise ValueError("kernel_prior must return at least 'k_length' and 'k_scale' (kernels.py:44-91)")
            if self.kernel_name == "Periodic" and "period" not in have:
                raise ValueError("kernel_prior of a Periodic kernel must return 'period' (kernels.py:94-117)")
            for sx in sites:
                if sx.name == "k_length" and sx.size not in (1, self.kernel_dim):
                    raise ValueError(f"k_length site has {sx.size} entries for input_dim {self.kernel_dim}")
            self._det.update({k: v for k, v in det.items() if k in ("k_length", "k_scale", "period")})
            return sites
        # the default priors are the program the reference runs (gp.py:229-247), traced like a user's kernel_prior: a
        # subclass that overrides _sample_kernel_params — as the reference's own subclasses do — changes the sites
        sites, returned, det = self._traced(self._sample_kernel_params, "_sample_kernel_params")
        self._det.update({k: v for k, v in det.items() if k in ("k_length", "k_scale", "period")})
        return sites

    def _sample_noise(self):
        """The reference's default noise prior as a program (gp.py:222-227): `sample("noise", noise_prior_dist or
        LogNormal(0, 1))` — meaningful inside a trace (models trace it once to learn the site); override to change it."""
        noise_dist = self.noise_prior_dist if self.noise_prior_dist is not None else dist.LogNormal(0.0, 1.0)
        return sample("noise", noise_dist)

    def _sample_kernel_params(self, output_scale=True) -> Dict[str, np.ndarray]:
        """The reference's default kernel priors as a program (gp.py:229-247): k_length under plate 'ard'
        (lengthscale_prior_dist or LogNormal(0, 1)), k_scale LogNormal(0, 1) — or the constant 1 when not output_scale —
        and, for the Periodic kernel, period LogNormal(0, 1)."""
        length_dist = self.lengthscale_prior_dist if self.lengthscale_prior_dist is not None else dist.LogNormal(0.0, 1.0)
        with plate("ard", self.kernel_dim):  # gp.py:238-239
            length = sample("k_length", length_dist)
        if output_scale:
            scale = sample("k_scale", dist.LogNormal(0.0, 1.0))  # gp.py:241
        else:
            scale = deterministic("k_scale", 1.0)
        period = sample("period", dist.LogNormal(0.0, 1.0)) if self.kernel_name == "Periodic" else None  # gp.py:243-244
        return {"k_length": length, "k_scale": scale, "period": period}

    def _noise_sites(self):
        if self.noise_prior is not None:  # gp.py:146-147 (deprecated there)
            sites, returned, det = self._traced(self.noise_prior, "noise_prior")
            if len(sites) != 1 or sites[0].name != "noise" or sites[0].size != 1:
                raise ValueError("noise_prior must sample exactly one scalar site named 'noise'")
            return sites
        sites, _, _ = self._traced(self._sample_noise, "_sample_noise")  # gp.py:222-227
        if len(sites) != 1 or sites[0].name != "noise" or sites[0].size != 1:
            raise ValueError("_sample_noise must sample exactly one scalar site named 'noise'")
        return sites

    def _sites(self):
        return self._kernel_sites() + self._noise_sites() + self._mean_prior_sites()

    def _with_deterministic(self, theta):
        """numpyro.deterministic values registered by a kernel_prior (e.g. a fixed k_scale)."""
        for k, v in getattr(self, "_det", {}).items():
            theta.setdefault(k, v)
        return theta

    _chains_over_devices = True  # _log_joint_batch honours its `eng` argument (subclasses that do not set this False)

    def _chain_devices(self, chain_method: str, device, num_chains: int):
        """GPU ordinals the chains of a concurrent NUTS run are dealt over: `device` "all" / a list, or — for
        chain_method='parallel' with no device given — every visible GPU; one device otherwise."""
        if not self._chains_over_devices or not isinstance(_lib.get_engine(self._device), _lib.Engine):
            return [None]
        if device == "all" or (device is None and chain_method == "parallel"):
            devs = list(range(_lib.visible_device_count()))
        elif isinstance(device, (list, tuple)):
            devs = [int(v) for v in device]
        else:
            return [None]
        devs = devs[:num_chains]
        return devs if len(devs) > 1 else [None]

    def _engine_on(self, dev: int) -> _lib.Engine:
        """A context of its own on GPU `dev` holding this model's training set (one per group of chains)."""
        eng = _lib.Engine(int(dev))
        self._prepare_engine(eng)
        return eng

    def _prepare_engine(self, eng) -> None:
        eng.set_train(self.X_train)

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

    def _dmean(self, X, theta, name) -> np.ndarray:
        """Jacobian d mean_fn(X, theta) / d theta[name], one row per element of the site (size, N): the user's
        mean_fn_grad, else complex-step differentiation (f(x + ih).imag / h, h = 1e-30: no subtractive cancellation,
        exact to rounding for analytic NumPy expressions), else central differences — element by element, so a
        vector-valued site (a plate inside mean_fn_prior) gets its full Jacobian.  N here is the flattened size of
        mean_fn's output ((T, N) -> T N for the task-batched models)."""
        base = np.asarray(theta[name], dtype=np.float64)
        size = max(1, base.size)
        if self.mean_fn_grad is not None:
            return np.asarray(self.mean_fn_grad(X, theta)[name], dtype=np.float64).reshape(size, -1)

        def with_element(i, value, dtype):
            if base.ndim == 0:
                return value
            arr = base.astype(dtype)
            arr.reshape(-1)[i] = value
            return arr

        mshape = np.shape(self._mean(X, theta))  # (N,), or (T, N) for the task-batched models
        J = np.empty((size, int(np.prod(mshape))))
        for i in range(size):
            x0 = float(base.reshape(-1)[i])
            row = None
            try:
                tc = dict(theta)
                tc[name] = with_element(i, complex(x0, 1e-30), np.complex128)
                with np.errstate(all="ignore"):
                    out = np.asarray(self.mean_fn(X.astype(np.complex128), tc))
                if np.iscomplexobj(out):
                    dm = out.imag.squeeze() / 1e-30
                    if dm.shape == mshape and np.all(np.isfinite(dm)):
                        row = dm
            except Exception:
                row = None
            if row is None:
                h = 1e-6 * max(1.0, abs(x0))
                tp, tm = dict(theta), dict(theta)
                tp[name], tm[name] = with_element(i, x0 + h, np.float64), with_element(i, x0 - h, np.float64)
                row = (self._mean(X, tp) - self._mean(X, tm)) / (2 * h)
            J[i] = np.asarray(row, dtype=np.float64).reshape(-1)
        return J

    def _unpack(self, sites, u):
        theta, off = {}, 0
        for s in sites:
            ui = u[off:off + s.size]
            x = s.dist.transform(ui)
            theta[s.name] = x.reshape(s.shape) if s.shape else float(x[0])
            off += s.size
        return self._with_deterministic(theta)

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
        if plan is not None and (glik is None or all(s.name in glik and np.size(glik[s.name]) == s.size for s in sites)):
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
                        if gx.size != s.size:
                            # a site with fewer values than the device gradient has entries: one lengthscale shared by all
                            # input dimensions (custom kernel_prior) — per TASK when the site carries the task axis
                            # (vExactGP: gradient (T, d), site (T,) or (T, 1)): each site element collects its own row
                            if s.size > 1 and gx.size % s.size == 0:
                                gx = gx.reshape(s.size, -1).sum(axis=1)
                            else:
                                gx = np.array([gx.sum()])
                    else:  # mean-function parameter: d lml / d phi = sum_i alpha_i d m_i / d phi
                        gx = self._dmean(self.X_train, theta, s.name) @ np.asarray(alpha, dtype=np.float64).reshape(-1)
                    gu = (gx + d_.grad_log_prob(x)) * d_.dx_du(ui)
                    if jacobian:
                        gu = gu + dlj
                    grad[off:off + s.size] = gu
                off += s.size
        return val, grad

    def _native_transition(self, sites, jitter: float, rng):
        """The NUTS transition as ONE call into the library (gpx_nuts_transition, csrc/nuts.hip: the loop of infer/nuts.py in
        C++ around the device fit step) for the default model — every site LogNormal (gp.py:222-247), no mean function, no
        subclass hooks, a real device context, a PCG64 generator — else None: the Python loop.  Same chain either way (the
        library draws its uniforms from `rng`'s own stream); GPX_NATIVE_NUTS=0 keeps the Python loop."""
        import os
        if os.environ.get("GPX_NATIVE_NUTS", "1") == "0" or self.mean_fn is not None or getattr(self, "_det", None):
            return None
        cls = type(self)
        if any(getattr(cls, name) is not getattr(ExactGP, name)
               for name in ("_log_joint", "_log_joint_batch", "_chain_rule", "_prepare_engine", "_unpack", "_ell")):
            return None
        plan = self._lognormal_plan(sites)
        want = {"k_length": self.kernel_dim, "k_scale": 1, "noise": 1}
        if self.kernel_name == "Periodic":
            want["period"] = 1
        if plan is None or {s.name: s.size for s in sites} != want or len(sites) != len(want):
            return None
        if type(rng.bit_generator).__name__ != "PCG64":
            return None
        eng = self._engine()
        if not isinstance(eng, _lib.Engine) or self.kernel_dim + 3 > 20:
            return None
        off, at = {}, 0
        for s_ in sites:
            off[s_.name] = at
            at += s_.size
        idx_ell = [off["k_length"] + c for c in range(self.kernel_dim)] + ([off["period"]] if "period" in off else [])
        nplan = eng.nuts_plan(self._kind, idx_ell, off["k_scale"], off["noise"], plan[0], plan[1], plan[2], jitter, self.y_train)

        def transition(u, U, g, eps, inv_mass, rng_, max_tree_depth):
            p0 = rng_.standard_normal(u.shape[0]) / np.sqrt(inv_mass)
            return self._engine().nuts_transition(nplan, u, U, g, p0, eps, inv_mass, max_tree_depth, rng_)

        transition.plan = nplan  # (tests: the same potential through gpx_nuts_potential)
        return transition

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
        # chain_method='parallel' places the chains on separate devices in the reference (NumPyro pmaps them over
        # jax.local_devices(), gp.py:173-174,214): here chain c runs on GPU devices[c % G], the chains of one GPU
        # advancing in lockstep as one batched device pass; 'vectorized' keeps every chain on one GPU.  Chains are
        # independent and each chain's arithmetic does not depend on its batch, so the draws equal the sequential ones.
        devices = self._chain_devices(chain_method, device, num_chains) if concurrent else [None]
        locksteps, group_engines = {}, []
        if concurrent:
            for g_, dev in enumerate(devices):
                members = [c for c in range(num_chains) if c % len(devices) == g_]
                eng_g = None if len(devices) == 1 else self._engine_on(dev)
                if eng_g is not None:
                    group_engines.append(eng_g)
                ls = _Lockstep(len(members),
                               lambda us, e=eng_g: self._log_joint_batch(sites, us, jitter, jacobian=True, eng=e))
                for c in members:
                    locksteps[c] = ls

        def run_chain(c):
            lockstep = locksteps.get(c)
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
                # one chain (the reference's default, and what every notebook runs): the transition loop inside the library.
                # Several chains keep the Python loop whatever the chain_method, so that 'parallel' / 'vectorized' chains —
                # which advance in lockstep through _log_joint_batch — stay bit for bit the 'sequential' ones.
                native = self._native_transition(sites, jitter, crng) if num_chains == 1 else None
                results[c] = run_nuts(potential, u0, num_warmup, num_samples, crng, progress=prog, transition=native)
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
            for e_ in group_engines:
                e_.close()
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
        # numpyro.deterministic sites of a kernel_prior (constants, e.g. a fixed period) are part of MCMC.get_samples()
        for name, val in getattr(self, "_det", {}).items():
            if name not in samples:
                v = np.asarray(val, dtype=np.float64)
                samples[name] = np.broadcast_to(v, draws.shape[:2] + v.shape).copy()
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
        `device`: a GPU ordinal, or "all" / a list of ordinals to shard the posterior samples over the GPUs of
        this node (one process, RCCL over xGMI; the reference's `device` is a jax.Device for device_put).
        """
        X_new = self._set_data(X_new)
        if samples is None:
            samples = self.get_samples(chain_dim=False)
        if isinstance(device, int):
            self._device = device
        jitter = float(kwargs.get("jitter", 1e-6))
        m_slice = int(kwargs.pop("_m_slice", 0))  # predict_in_batches: covariance blocks of this many test points
        ells, scales, noises, yres, eps, mean_shift = self._sweep_inputs(rng_key, X_new, samples, n, m_slice)
        if device == "all" or isinstance(device, (list, tuple)):
            # the vmap axis sharded over the GPUs of this node from this one process: RCCL broadcast of the inputs,
            # RCCL gather of the results (gpx_predict_sweep_multi); same values as the single-GPU sweep
            node = _lib.get_node(None if device == "all" else list(device))
            means, y_sampled, infos = node.predict_sweep(self.X_train, self._kind, ells, scales, noises, yres, X_new,
                                                         noiseless, jitter, eps, m_slice=m_slice)
            return self._sweep_outputs(means, y_sampled, mean_shift, filter_nans)
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
                            n: int = 1, filter_nans: bool = False, noiseless: bool = False, comm=None,
                            **kwargs: float) -> Optional[Tuple[np.ndarray, np.ndarray]]:
        """predict() with the posterior samples sharded over the processes of a one-process-per-GPU launch
        (`python -m torch.distributed.run`, mpirun, gpax_amd.launch.spawn_ranks: RANK / LOCAL_RANK / WORLD_SIZE).
        A collective call: every rank enters it with a model of the same configuration; only rank 0 needs the
        fitted state / arguments.  Rank 0 returns (y_mean, y_sampled), the other ranks None (gp.py:351-399's vmap
        axis is the only axis of the path that shards).
        comm: None — the process-wide `_lib.Rank` of this launch (gpax_amd.launch.default_rank(): the library's own
        RCCL communicator, ncclBroadcast of the inputs and ncclSend / ncclRecv gather of the results over xGMI; no
        other runtime involved); a `_lib.Rank`; or any object with the protocol of gpax_amd.parallel (rank, world,
        bcast, gather_rows) for callers living in another runtime's process group."""
        from .. import launch
        jitter = float(kwargs.get("jitter", 1e-6))
        if comm is None:
            comm = launch.default_rank()
        root = comm.rank == 0
        mean_shift = None
        ells = scales = noises = yres = eps = None
        if root:
            X_new = self._set_data(X_new)
            if samples is None:
                samples = self.get_samples(chain_dim=False)
            ells, scales, noises, yres, eps, mean_shift = self._sweep_inputs(rng_key, X_new, samples, n)
        if isinstance(comm, _lib.Rank):
            # sizes travel first (16 doubles), then ONE collective sweep inside the library
            hdr = np.zeros(16)
            if root:
                yres2 = np.atleast_2d(yres)
                hdr[:9] = [self.X_train.shape[0], self.X_train.shape[1], ells.shape[0], X_new.shape[0], n,
                           self._kind, 1.0 if noiseless else 0.0, jitter, yres2.shape[0]]
            hdr = comm.bcast(hdr)
            N, d, S, M, n_, kind, nl, jit, rows = (int(hdr[0]), int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4]),
                                                   int(hdr[5]), bool(hdr[6]), float(hdr[7]), int(hdr[8]))
            res = comm.predict_sweep(kind, N, d, S, M, n_, nl, jit, yres_rows=rows, X=self.X_train if root else None,
                                     ells=ells, scales=scales, noises=noises, yres=yres, Xnew=X_new if root else None,
                                     eps=eps)
        else:
            from ..parallel import predict_sharded
            args = (self.X_train, yres, X_new, {"k_length": ells, "k_scale": scales, "noise": noises}, eps) if root \
                else (None, None, None, None, None)
            res = predict_sharded(_lib.get_sweep_engines(self._device), self._kind, *args, noiseless, jitter, comm)
        if res is None:
            return None
        means, y_sampled, infos = res
        return self._sweep_outputs(means, y_sampled, mean_shift, filter_nans)

    def sample_from_prior(self, rng_key, X: np.ndarray, num_samples: int = 10, **kwargs: float):
        """Samples from the prior predictive distribution at X (gp.py:401-408: Predictive(self.model, num_samples)):
        theta ~ priors, y ~ MVN(mean_fn(X), kernel(X, X, theta, noise, jitter)).  Randomness is consumed per draw as
        [one draw of every site, in site order; then N standard normals] from rng_from_key(rng_key); the Gram matrix
        and its Cholesky factor come from the device."""
        X = self._set_data(X)
        rng = rng_from_key(rng_key)
        jitter = float(kwargs.get("jitter", 1e-6))
        eng = _lib.get_engine(self._device)
        sites = self._sites()
        out = np.empty((num_samples, X.shape[0]))
        for i in range(num_samples):
            theta = self._with_deterministic(
                {s.name: (s.dist.sample(rng, s.shape) if s.shape else float(s.dist.sample(rng))) for s in sites})
            eps = rng.standard_normal(X.shape[0])
            K = eng.gram(self._kind, X, X, self._ell(theta), self._scalar(theta["k_scale"]),
                         self._scalar(theta["noise"]) + jitter, True)
            L, info = eng.potrf(K)
            out[i] = self._mean(X, theta) + L @ eps if info == 0 else np.nan
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
