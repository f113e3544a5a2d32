"""
Host-side helpers mirroring gpax/utils/utils.py (the seven on the exact-GP path).

PRNG keys: the reference passes JAX threefry keys; those streams cannot be reproduced without
JAX, so a "key" here is an opaque seed object — get_keys(seed) returns two uint32[2] arrays
derived with numpy.random.SeedSequence, and every consumer turns a key into a
numpy.random.Generator with `rng_from_key` (ints, Generators and any array-like are accepted).
"""
from typing import Dict, List, Union

import numpy as np


def enable_x64():
    """gpax/utils/utils.py:19-21.  The MI355X path is fp64 end to end; kept for API parity."""
    return None


def get_keys(seed: int = 0):
    """gpax/utils/utils.py:24-30: two independent keys for inference and prediction."""
    ss = np.random.SeedSequence(int(seed))
    k1, k2 = ss.spawn(2)
    return k1.generate_state(2, dtype=np.uint32), k2.generate_state(2, dtype=np.uint32)


def rng_from_key(rng_key) -> np.random.Generator:
    if isinstance(rng_key, np.random.Generator):
        return rng_key
    if hasattr(rng_key, "k") and type(rng_key).__name__ == "ThreefryKey":
        # host-side consumers other than predict's normals (NUTS, SVI, init): a Generator seeded from the key words
        return np.random.default_rng([int(rng_key.k[0]), int(rng_key.k[1])])
    if rng_key is None:
        return np.random.default_rng()
    arr = np.asarray(rng_key)
    if arr.ndim == 0:
        return np.random.default_rng(int(arr) & 0xFFFFFFFFFFFFFFFF)
    return np.random.default_rng([int(v) & 0xFFFFFFFF for v in arr.reshape(-1)])


def split_key(rng_key, num: int = 2):
    """jax.random.split analogue on opaque keys."""
    rng = rng_from_key(rng_key)
    return [rng.integers(0, 2 ** 32, size=2, dtype=np.uint32) for _ in range(num)]


def split_in_batches(X_new: np.ndarray, batch_size: int = 100, dim: int = 0):
    """gpax/utils/utils.py:33-51.  The reference raises UnboundLocalError when the array is
    shorter than one batch (its loop variable is never bound); here that case returns the
    single short batch."""
    if dim not in [0, 1]:
        raise NotImplementedError("'dim' must be equal to 0 or 1")
    n = X_new.shape[dim]
    num_batches = n // batch_size
    X_split = []
    for i in range(num_batches):
        X_i = X_new[i * batch_size:(i + 1) * batch_size] if dim == 0 else X_new[:, i * batch_size:(i + 1) * batch_size]
        X_split.append(X_i)
    rest = num_batches * batch_size
    X_i = X_new[rest:] if dim == 0 else X_new[:, rest:]
    if X_i.shape[dim] > 0:
        X_split.append(X_i)
    return X_split


def split_dict(data: Dict[str, np.ndarray], chunk_size: int) -> List[Dict[str, np.ndarray]]:
    """Cut every array of a samples dictionary into consecutive pieces of `chunk_size` rows (the last one may be
    shorter) — one dictionary per piece.  Behaviour of gpax/utils/utils.py:54-81."""
    rows = len(next(iter(data.values())))
    cuts = range(0, rows, int(chunk_size))
    return [{name: arr[lo:lo + chunk_size] for name, arr in data.items()} for lo in cuts]


def random_sample_dict(data: Dict[str, np.ndarray], num_samples: int, rng_key) -> Dict[str, np.ndarray]:
    """gpax/utils/utils.py:84-102: the same random subset of every array."""
    num_data_points = len(next(iter(data.values())))
    indices = rng_from_key(rng_key).permutation(num_data_points)[:num_samples]
    return {key: value[indices] for key, value in data.items()}


def preprocess_sparse_image(sparse_image: np.ndarray):
    """Zeros mark missing pixels (behaviour of gpax/utils/utils.py:150-168).  Returns, in the image dtype:
    X_train (N, D) — coordinates of the measured pixels in row-major order; y (N,) — their values;
    X_full (prod(shape), D) — every pixel coordinate.  For images (D = 2) X_full is row-major too, so a prediction on
    it reshapes straight back to the image; for D >= 3 the reference's transposed 'xy' meshgrid enumerates the axes
    slow -> fast as D-1, ..., 2, 0, 1, and that order is kept."""
    img = np.asarray(sparse_image)
    measured = img != 0
    coords = np.argwhere(measured)
    slow_to_fast = list(range(img.ndim - 1, 1, -1)) + list(range(min(img.ndim, 2)))
    counters = np.indices([img.shape[ax] for ax in slow_to_fast]).reshape(img.ndim, -1)
    grid = np.empty((counters.shape[1], img.ndim), dtype=img.dtype)
    for pos, ax in enumerate(slow_to_fast):
        grid[:, ax] = counters[pos]
    return coords.astype(img.dtype), img[measured].astype(img.dtype), grid


def initialize_inducing_points(X, ratio=0.1, method='uniform', key=None):
    """int(N * ratio) inducing inputs out of X (behaviour of gpax/utils/utils.py:171-212): 'uniform' = evenly spaced
    rows (the reference's int8 index dtype overflows beyond 128 points, utils.py:191; full-width integers here),
    'random' = a draw without replacement (needs `key`), 'kmeans' = cluster centres (scikit-learn)."""
    X = np.asarray(X)
    if not (0 < ratio < 1):
        raise ValueError("The 'ratio' value must be between 0 and 1")
    count = int(X.shape[0] * ratio)

    def evenly_spaced():
        return X[np.linspace(0, X.shape[0] - 1, count).astype(np.int64)]

    def random_rows():
        if key is None:
            raise ValueError("A JAX random key must be provided for random selection")
        return X[rng_from_key(key).choice(X.shape[0], size=(count,), replace=False)]

    def cluster_centres():
        try:
            from sklearn.cluster import KMeans
        except ImportError as e:  # pragma: no cover
            raise ImportError("You need to install `scikit-learn` to be able to use this feature.") from e
        return np.asarray(KMeans(n_clusters=count, random_state=0, n_init=10).fit(X).cluster_centers_)

    pick = {'uniform': evenly_spaced, 'random': random_rows, 'kmeans': cluster_centres}.get(method)
    if pick is None:
        raise ValueError("Method must be 'uniform', 'random', or 'kmeans'")
    return pick()


def set_fn(func):
    """gpax/utils/fn.py:21-55: turn a deterministic function f(x, a, b, ...) into f(x, params) reading its
    parameters from a dictionary.  (The reference rewrites the source; a keyword-forwarding wrapper has the same
    behaviour.  set_kernel_fn — custom kernels — has no MI355X path.)"""
    import inspect

    names = list(inspect.signature(func).parameters.keys())[1:]

    def wrapped(x, params):
        return func(x, **{n: params[n] for n in names})

    wrapped.__name__ = getattr(func, "__name__", "fn")
    wrapped.__doc__ = func.__doc__
    return wrapped
