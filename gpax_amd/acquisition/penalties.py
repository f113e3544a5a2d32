"""Recent-point penalties for the acquisition functions (gpax/acquisition/penalties.py:6-66)."""
import numpy as np


def compute_penalty(X: np.ndarray, recent_points: np.ndarray, penalty_type: str = "delta",
                    penalty_factor: float = 1.0) -> np.ndarray:
    """penalties.py:6-34"""
    if penalty_type not in ["delta", "inverse_distance", "inverse distance"]:
        raise NotImplementedError("Avaialble penalty types are 'delta' and 'inverse distance'")
    X = np.asarray(X, dtype=np.float64)
    recent_points = np.asarray(recent_points, dtype=np.float64)
    if penalty_type == "delta":
        return find_and_replace_point_indices(X, recent_points)
    return penalty_factor * np.array([penalty_point(x, recent_points) for x in X])


def penalty_point(x: np.ndarray, recent_points: np.ndarray) -> float:
    """penalties.py:37-50: sum over recent points of 1 / (distance + 1) / age."""
    if recent_points.ndim == 1:
        recent_points = recent_points[:, None]
    distances = np.linalg.norm(recent_points - x, axis=1)
    if len(recent_points) == 1:
        timestamps = 1
    else:
        timestamps = np.arange(len(recent_points) + 1, 1, -1)
    return float(np.sum(1 / (distances + 1) / timestamps))


def find_and_replace_point_indices(points: np.ndarray, other_points: np.ndarray) -> np.ndarray:
    """penalties.py:53-66: +inf at the first grid point equal to each recent point."""
    out = np.zeros(len(points))
    for single_point in other_points:
        index = np.where(np.all(points == single_point, axis=1))
        if index[0].size > 0:
            out[index[0][0]] = np.inf
    return out
