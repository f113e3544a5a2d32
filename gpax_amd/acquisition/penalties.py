"""Recent-point penalties for the acquisition functions (gpax/acquisition/penalties.py:6-66), as whole-array NumPy
operations: one (M, R) distance / equality table per call instead of a Python loop over the M candidate points."""
import numpy as np

_TYPES = ("delta", "inverse_distance", "inverse distance")


def _as_points(a, d=None) -> np.ndarray:
    """(R, d) view of a set of points; a 1-D array is R scalars (penalties.py:41-42) unless d says it is ONE point."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[None, :] if (d is not None and d > 1 and a.shape[0] == d) else a[:, None]
    return a


def _age_weights(R: int) -> np.ndarray:
    """1 / timestamp: the most recent point (last row) weighs 1/2, the oldest 1/(R + 1); a single point weighs 1
    (penalties.py:45-48)."""
    return np.ones(1) if R == 1 else 1.0 / np.arange(R + 1, 1, -1, dtype=np.float64)


def penalty_point(x: np.ndarray, recent_points: np.ndarray) -> float:
    """penalties.py:37-50: sum over the recent points of 1 / (distance + 1) / age, for ONE candidate x."""
    rp = _as_points(recent_points)
    dist = np.sqrt(np.sum((rp - np.asarray(x, dtype=np.float64)) ** 2, axis=1))
    return float(np.sum(_age_weights(len(rp)) / (dist + 1.0)))


def find_and_replace_point_indices(points: np.ndarray, other_points: np.ndarray) -> np.ndarray:
    """penalties.py:53-66: +inf at the FIRST candidate equal to each recent point, 0 elsewhere."""
    points = np.asarray(points)
    out = np.zeros(len(points))
    others = np.asarray(other_points)
    if others.size == 0:
        return out
    if others.ndim == 1:  # R scalars, each compared with every coordinate (the reference's broadcast)
        others = others[:, None]
    eq = np.all(points[:, None, :] == others[None, :, :], axis=2)  # (M, R)
    hit = eq.any(axis=0)
    out[eq.argmax(axis=0)[hit]] = np.inf
    return out


def compute_penalty(X: np.ndarray, recent_points: np.ndarray, penalty_type: str = "delta",
                    penalty_factor: float = 1.0) -> np.ndarray:
    """penalties.py:6-34: 'delta' marks the recent points themselves, 'inverse_distance' penalises their neighbourhood."""
    if penalty_type not in _TYPES:
        raise NotImplementedError("Avaialble penalty types are 'delta' and 'inverse distance'")  # (sic, penalties.py:28)
    X = np.asarray(X, dtype=np.float64)
    recent_points = np.asarray(recent_points, dtype=np.float64)
    if penalty_type == "delta":
        return find_and_replace_point_indices(X, recent_points)
    rp = _as_points(recent_points)
    Xc = X if X.ndim > 1 else X[:, None]
    dist = np.sqrt(np.sum((Xc[:, None, :] - rp[None, :, :]) ** 2, axis=2))  # (M, R)
    return penalty_factor * ((1.0 / (dist + 1.0)) @ _age_weights(len(rp)))
