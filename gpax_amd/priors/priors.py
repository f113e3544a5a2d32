"""
Prior helpers with the reference's names (gpax/priors/priors.py:71-267): distribution constructors to pass as
`lengthscale_prior_dist` / `noise_prior_dist`, and automatic priors over the parameters of a deterministic
mean function.

The reference's auto_*_priors return a callable that runs numpyro.sample per parameter; models here take
`mean_fn_prior` as a dict name -> distribution, so the auto_* helpers return that dict (same parameter
discovery: the function's signature minus its leading arguments).  The place_*_prior functions are the reference's
(priors.py:18-68) with the import swapped — `gpax_amd.sample` instead of `numpyro.sample` — for use inside a prior
callable a model traces (mean_fn_prior / kernel_prior / noise_prior).  The *_kernel_priors variants only exist to build
NumPyro programs for custom kernels, which have no MI355X path.
"""
from __future__ import annotations

import inspect
from typing import Callable, Dict

import numpy as np

from ..infer import dist


def normal_dist(loc: float = None, scale: float = None) -> dist.Distribution:
    """Normal(loc, scale), defaults 0 and 1 (priors.py:71-90)."""
    return dist.Normal(0.0 if loc is None else loc, 1.0 if scale is None else scale)


def lognormal_dist(loc: float = None, scale: float = None) -> dist.Distribution:
    """LogNormal(loc, scale), defaults 0 and 1 (priors.py:93-111)."""
    return dist.LogNormal(0.0 if loc is None else loc, 1.0 if scale is None else scale)


def halfnormal_dist(scale: float = None) -> dist.Distribution:
    """HalfNormal(scale), default 1 (priors.py:114-131)."""
    return dist.HalfNormal(1.0 if scale is None else scale)


def gamma_dist(c: float = None, r: float = None, input_vec: np.ndarray = None) -> dist.Distribution:
    """Gamma(c, r); a missing shape c is inferred as half the range of input_vec, r defaults to 1
    (priors.py:134-161)."""
    if c is None:
        if input_vec is not None:
            input_vec = np.asarray(input_vec, dtype=np.float64)
            c = (input_vec.max() - input_vec.min()) / 2
        else:
            raise ValueError("Provide either c or an input array")
    if r is None:
        r = 1.0
    return dist.Gamma(c, r)


def uniform_dist(low: float = None, high: float = None, input_vec: np.ndarray = None) -> dist.Distribution:
    """Uniform(low, high); missing bounds are taken from the min / max of input_vec (priors.py:164-189)."""
    if (low is None or high is None) and input_vec is None:
        raise ValueError("If 'low' or 'high' is not provided, an input array must be provided.")
    if input_vec is not None:
        input_vec = np.asarray(input_vec, dtype=np.float64)
    low = low if low is not None else input_vec.min()
    high = high if high is not None else input_vec.max()
    return dist.Uniform(low, high)


def place_normal_prior(param_name: str, loc: float = 0.0, scale: float = 1.0):
    """sample(param_name, Normal(loc, scale)) (priors.py:18-24) — inside a prior callable passed to a model."""
    from ..infer.primitives import sample
    return sample(param_name, normal_dist(loc, scale))


def place_lognormal_prior(param_name: str, loc: float = 0.0, scale: float = 1.0):
    """sample(param_name, LogNormal(loc, scale)) (priors.py:27-33)."""
    from ..infer.primitives import sample
    return sample(param_name, lognormal_dist(loc, scale))


def place_halfnormal_prior(param_name: str, scale: float = 1.0):
    """sample(param_name, HalfNormal(scale)) (priors.py:36-42)."""
    from ..infer.primitives import sample
    return sample(param_name, halfnormal_dist(scale))


def place_uniform_prior(param_name: str, low: float = None, high: float = None, X: np.ndarray = None):
    """sample(param_name, Uniform(low, high)), missing bounds from the range of X (priors.py:45-55)."""
    from ..infer.primitives import sample
    return sample(param_name, uniform_dist(low, high, X))


def place_gamma_prior(param_name: str, c: float = None, r: float = None, X: np.ndarray = None):
    """sample(param_name, Gamma(c, r)), a missing shape from the range of X (priors.py:58-68)."""
    from ..infer.primitives import sample
    return sample(param_name, gamma_dist(c, r, X))


def auto_priors(func: Callable, params_begin_with: int, dist_type: str = 'normal', loc: float = 0.0,
                scale: float = 1.0) -> Dict[str, dist.Distribution]:
    """A Normal / LogNormal(loc, scale) prior for every parameter of `func` from position
    `params_begin_with` on (priors.py:192-216), as the dict the models take for `mean_fn_prior`."""
    make = dist.LogNormal if dist_type == 'lognormal' else dist.Normal
    names = list(inspect.signature(func).parameters.keys())[params_begin_with:]
    return {name: make(loc, scale) for name in names}


def auto_normal_priors(func: Callable, loc: float = 0.0, scale: float = 1.0) -> Dict[str, dist.Distribution]:
    """Normal priors over the parameters of a mean function f(x, a, b, ...) (priors.py:219-232)."""
    return auto_priors(func, 1, 'normal', loc, scale)


def auto_lognormal_priors(func: Callable, loc: float = 0.0, scale: float = 1.0) -> Dict[str, dist.Distribution]:
    """LogNormal priors over the parameters of a mean function f(x, a, b, ...) (priors.py:235-248)."""
    return auto_priors(func, 1, 'lognormal', loc, scale)
