"""Synthetic workloads of BASELINE.md §3 — shared by bench.py, the tests and the oracle's callers.
Pure input generation (NumPy only): no part of the checker and no part of the product."""
import math

import numpy as np


def synthetic_problem(N: int, d: int, M: int, seed: int = 0, noise: float = 0.1):
    """X ~ U(0, 10)^d, y = prod_j sin(x_j + 0.3 j) + sqrt(noise) N(0, 1), X_new ~ U(0, 10)^d and the base
    hyper-parameters (k_length_j = 1 + 0.25 j, k_scale = 1.3, noise)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0.0, 10.0, size=(N, d))
    f = np.prod(np.sin(X + 0.3 * np.arange(d)[None, :]), axis=1)
    y = f + math.sqrt(noise) * rng.standard_normal(N)
    Xnew = rng.uniform(0.0, 10.0, size=(M, d))
    params = {"k_length": 1.0 + 0.25 * np.arange(d), "k_scale": 1.3, "noise": noise}
    return X, y, Xnew, params


def synthetic_theta_samples(S: int, d: int, seed: int = 1, noise: float = 0.1):
    """C4 of BASELINE.md: lengthscales / scale ~ LogNormal(0, 0.1) * base, noise ~ LogNormal(log .1, .1)."""
    rng = np.random.default_rng(seed)
    base_l = 1.0 + 0.25 * np.arange(d)
    return {
        "k_length": base_l[None, :] * np.exp(0.1 * rng.standard_normal((S, d))),
        "k_scale": 1.3 * np.exp(0.1 * rng.standard_normal(S)),
        "noise": noise * np.exp(0.1 * rng.standard_normal(S)),
    }


def synthetic_sparse_image(H: int = 512, W: int = 512, keep: float = 0.0625, seed: int = 3):
    """C5 of BASELINE.md: an H x W image (sum of 6 Gaussians + 2 sinusoids), a random `keep` fraction of its
    pixels kept and the rest zeroed (zeros = missing, gpax/utils/utils.py:150-168).  Returns (image, sparse)."""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    img = np.full((H, W), 1.5)
    for _ in range(6):
        ci, cj = rng.uniform(0, H), rng.uniform(0, W)
        s = rng.uniform(0.06, 0.15) * min(H, W)
        img += rng.uniform(0.5, 1.0) * np.exp(-((ii - ci) ** 2 + (jj - cj) ** 2) / (2 * s * s))
    img += 0.3 * np.sin(ii / (0.08 * H)) * np.cos(jj / (0.11 * W)) + 0.2 * np.sin((ii + jj) / (0.1 * (H + W)))
    mask = rng.uniform(size=img.shape) < keep
    return img, np.where(mask, img, 0.0)
