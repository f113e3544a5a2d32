"""
Closed-form acquisition functions on the predictive moments (gpax/acquisition/base_acq.py:20-155).
Element-wise O(M) work on the (mean, variance) vectors the GPU pipeline returns — host NumPy.
("next" row 1 of SURVEY.md §8f: the immediate consumers of predict / get_mvn_posterior.)
"""
from typing import Tuple

import numpy as np
from scipy.special import ndtr

_SQRT_2PI = 2.5066282746310002


def _pdf(u):
    return np.exp(-0.5 * u * u) / _SQRT_2PI


def ei(moments: Tuple[np.ndarray, np.ndarray], best_f: float = None, maximize: bool = False, **kwargs) -> np.ndarray:
    """Expected improvement (base_acq.py:20-73): sigma (phi(u) + u Phi(u)), u = +-(mean - best_f)/sigma."""
    mean, var = np.asarray(moments[0], dtype=np.float64), np.asarray(moments[1], dtype=np.float64)
    if best_f is None:
        best_f = mean.max() if maximize else mean.min()
    sigma = np.sqrt(var)
    u = (mean - best_f) / sigma
    if not maximize:
        u = -u
    return sigma * (_pdf(u) + u * ndtr(u))


def ucb(moments: Tuple[np.ndarray, np.ndarray], beta: float = 0.25, maximize: bool = False, **kwargs) -> np.ndarray:
    """Upper confidence bound (base_acq.py:76-109); negated lower bound when minimising."""
    mean, var = np.asarray(moments[0], dtype=np.float64), np.asarray(moments[1], dtype=np.float64)
    delta = np.sqrt(beta * var)
    return mean + delta if maximize else -(mean - delta)


def ue(moments: Tuple[np.ndarray, np.ndarray], **kwargs) -> np.ndarray:
    """Uncertainty-based exploration (base_acq.py:112-133): the predictive standard deviation."""
    return np.sqrt(np.asarray(moments[1], dtype=np.float64))


def poi(moments: Tuple[np.ndarray, np.ndarray], best_f: float = None, xi: float = 0.01, maximize: bool = False,
        **kwargs) -> np.ndarray:
    """Probability of improvement (base_acq.py:136-155)."""
    mean, var = np.asarray(moments[0], dtype=np.float64), np.asarray(moments[1], dtype=np.float64)
    if best_f is None:
        best_f = mean.max() if maximize else mean.min()
    u = (mean - best_f - xi) / np.sqrt(var)
    if not maximize:
        u = -u
    return ndtr(u)
