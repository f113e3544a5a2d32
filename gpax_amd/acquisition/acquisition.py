"""
Acquisition-function wrappers with the reference signatures (gpax/acquisition/acquisition.py:22-524):
EI, UCB, POI, UE and Thompson on top of model.predict.  KG, the q-batch variants and optimize_acq
(jaxopt L-BFGS) are outside this build.
"""
from typing import Optional, Tuple

import numpy as np

from ..utils.utils import rng_from_key
from .base_acq import ei, poi, ucb, ue
from .penalties import compute_penalty


def _compute_mean_and_var(rng_key, model, X, n: int, noiseless: bool, **kwargs) -> Tuple[np.ndarray, np.ndarray]:
    """acquisition.py:22-35: HMC models -> moments of the pooled draws; VI models -> (mean, var)."""
    if getattr(model, "mcmc", None) is not None:
        _, y_sampled = model.predict(rng_key, X, n=n, noiseless=noiseless, **kwargs)
        y_sampled = y_sampled.reshape(n * y_sampled.shape[0], -1)
        return y_sampled.mean(0), y_sampled.var(0)
    return model.predict(rng_key, X, noiseless=noiseless, **kwargs)


def _compute_penalties(X, recent_points, penalty, penalty_factor, grid_indices) -> np.ndarray:
    X_ = grid_indices if grid_indices is not None else X
    return compute_penalty(X_, recent_points, penalty, penalty_factor)


def _wrap(acq_fn, rng_key, model, X, n, noiseless, penalty, recent_points, grid_indices, penalty_factor, kwargs,
          **acq_args):
    if penalty and not isinstance(recent_points, np.ndarray):
        raise ValueError("Please provide an array of recently visited points")
    X = np.asarray(X, dtype=np.float64)
    X = X[:, None] if X.ndim < 2 else X
    moments = _compute_mean_and_var(rng_key, model, X, n, noiseless, **kwargs)
    acq = acq_fn(moments, **acq_args)
    if penalty:
        acq = acq - _compute_penalties(X, recent_points, penalty, penalty_factor, grid_indices)
    return acq


def EI(rng_key, model, X, best_f: float = None, maximize: bool = False, n: int = 1, noiseless: bool = False,
       penalty: Optional[str] = None, recent_points: np.ndarray = None, grid_indices: np.ndarray = None,
       penalty_factor: float = 1.0, **kwargs) -> np.ndarray:
    """Expected Improvement (acquisition.py:49-140)."""
    return _wrap(ei, rng_key, model, X, n, noiseless, penalty, recent_points, grid_indices, penalty_factor, kwargs,
                 best_f=best_f, maximize=maximize)


def UCB(rng_key, model, X, beta: float = .25, maximize: bool = False, n: int = 1, noiseless: bool = False,
        penalty: Optional[str] = None, recent_points: np.ndarray = None, grid_indices: np.ndarray = None,
        penalty_factor: float = 1.0, **kwargs) -> np.ndarray:
    """Upper confidence bound (acquisition.py:143-224)."""
    return _wrap(ucb, rng_key, model, X, n, noiseless, penalty, recent_points, grid_indices, penalty_factor, kwargs,
                 beta=beta, maximize=maximize)


def POI(rng_key, model, X, best_f: float = None, xi: float = 0.01, maximize: bool = False, n: int = 1,
        noiseless: bool = False, penalty: Optional[str] = None, recent_points: np.ndarray = None,
        grid_indices: np.ndarray = None, penalty_factor: float = 1.0, **kwargs) -> np.ndarray:
    """Probability of Improvement (acquisition.py:227-311)."""
    return _wrap(poi, rng_key, model, X, n, noiseless, penalty, recent_points, grid_indices, penalty_factor, kwargs,
                 best_f=best_f, xi=xi, maximize=maximize)


def UE(rng_key, model, X, n: int = 1, noiseless: bool = False, penalty: Optional[str] = None,
       recent_points: np.ndarray = None, grid_indices: np.ndarray = None, penalty_factor: float = 1.0,
       **kwargs) -> np.ndarray:
    """Uncertainty-based exploration (acquisition.py:314-394)."""
    return _wrap(ue, rng_key, model, X, n, noiseless, penalty, recent_points, grid_indices, penalty_factor, kwargs)


def Thompson(rng_key, model, X, n: int = 1, noiseless: bool = False, **kwargs) -> np.ndarray:
    """Thompson sampling (acquisition.py:488-524): one posterior sample of the GP parameters (HMC) or the
    point estimate (VI), then one function draw from its predictive distribution."""
    X = np.asarray(X, dtype=np.float64)
    X = X[:, None] if X.ndim < 2 else X
    rng = rng_from_key(rng_key)
    if getattr(model, "mcmc", None) is not None:
        posterior_samples = model.get_samples()
        idx = rng.integers(0, len(posterior_samples["k_length"]), size=(1,))
        samples = {k: v[idx] for (k, v) in posterior_samples.items()}
        _, tsample = model.predict(rng, X, samples, n, noiseless=noiseless, **kwargs)
        if n > 1:
            tsample = tsample.mean(1).squeeze()
        return tsample
    # VI models: the reference only defines this branch for viDKL (sample_from_posterior); here the draw comes from
    # the MVN posterior at the point estimate, factored on the device: the exact models go through
    # ExactGP._predict (gpx_posterior + gpx_mvn_draw: chol(cov) and mean + L eps, gp.py:279-293); a model with its own
    # posterior (viSparseGP: Woodbury) hands its covariance to the device Cholesky (gpx_potrf)
    from ..models.gp import ExactGP
    params = model.get_samples()
    if type(model).get_mvn_posterior is ExactGP.get_mvn_posterior:
        _, draws = ExactGP._predict(model, rng, X, params, 1, noiseless, **kwargs)
        return draws
    mean, cov = model.get_mvn_posterior(X, params, noiseless, **kwargs)
    eps = rng.standard_normal(len(mean))
    L, info = model._engine().potrf(cov + 1e-12 * np.eye(len(mean)))
    return (mean + L @ eps)[None] if info == 0 else np.full((1, len(mean)), np.nan)
