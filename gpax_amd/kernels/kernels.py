"""
Kernel callables with the reference protocol  k(X, Z, params, noise=0, jitter=1e-6) -> (n, m)
(gpax/kernels/kernels.py:17,44-91), evaluated by the HIP Gram kernel (gpx_gram).
"""
from typing import Callable, Dict, Union

import numpy as np

from .. import _lib

kernel_fn_type = Callable[[np.ndarray, np.ndarray, Dict[str, np.ndarray], np.ndarray], np.ndarray]


def _sqrt(x, eps=1e-12):
    """gpax/kernels/kernels.py:20-21"""
    return np.sqrt(x + eps)


def add_jitter(x, jitter=1e-6):
    """gpax/kernels/kernels.py:24-25"""
    return x + jitter


def _as2d(X):
    X = np.asarray(X, dtype=np.float64)
    return X if X.ndim > 1 else X[:, None]


def _scalar(v, name):
    a = np.asarray(v, dtype=np.float64).reshape(-1)
    if a.size != 1:
        raise ValueError(f"{name} must be a scalar (got shape {np.shape(v)})")
    return float(a[0])


def square_scaled_distance(X, Z, lengthscale=1.0) -> np.ndarray:
    """Squared distance between X and Z scaled by the lengthscale (gpax/kernels/kernels.py:28-41), on the GPU: the r^2 the
    Gram kernels evaluate, by itself — the direct sum_k ((x_k - z_k) / ell_k)^2 (never negative; the reference's expansion
    ||x/l||^2 - 2 x.z/l^2 + ||z/l||^2 clipped at 0 agrees with it to rounding).  lengthscale: scalar or (d,)."""
    X, Z = _as2d(X), _as2d(Z)
    ell = _lib.broadcast_lengthscale(lengthscale, X.shape[1])
    return _lib.get_engine().gram(_lib.KIND_R2, X, Z, ell, 1.0, 0.0, False)


def _gram(kind: int, X, Z, params, noise, jitter):
    X, Z = _as2d(X), _as2d(Z)
    scale = _scalar(params["k_scale"], "k_scale")
    ell = _lib.pack_ell(kind, params["k_length"], X.shape[1], params.get("period"))
    # the reference adds (noise + jitter) * eye iff X.shape == Z.shape (kernels.py:63,89)
    add_diag = X.shape == Z.shape
    diag_add = add_jitter(_scalar(noise, "noise"), jitter) if add_diag else 0.0
    return _lib.get_engine().gram(kind, X, Z, ell, scale, diag_add, add_diag)


def RBFKernel(X, Z, params: Dict[str, np.ndarray], noise=0, jitter: float = 1e-6, **kwargs) -> np.ndarray:
    """Radial basis function kernel (gpax/kernels/kernels.py:44-65) on the GPU."""
    return _gram(_lib.KERNEL_KINDS["RBF"], X, Z, params, noise, jitter)


def MaternKernel(X, Z, params: Dict[str, np.ndarray], noise=0, jitter: float = 1e-6, **kwargs) -> np.ndarray:
    """Matern-5/2 kernel (gpax/kernels/kernels.py:68-91) on the GPU."""
    return _gram(_lib.KERNEL_KINDS["Matern"], X, Z, params, noise, jitter)


def PeriodicKernel(X, Z, params: Dict[str, np.ndarray], noise=0, jitter: float = 1e-6, **kwargs) -> np.ndarray:
    """Periodic kernel (gpax/kernels/kernels.py:94-117) on the GPU; params: k_length, k_scale, period."""
    return _gram(_lib.KERNEL_KINDS["Periodic"], X, Z, params, noise, jitter)


PeriodicKernel.gpx_name = "Periodic"
RBFKernel.gpx_name = "RBF"
MaternKernel.gpx_name = "Matern"


def get_kernel(kernel: Union[str, kernel_fn_type] = 'RBF', **kwargs):
    """gpax/kernels/kernels.py:227-241.  'NNGP' has no MI355X path."""
    kernel_book = {'RBF': RBFKernel, 'Matern': MaternKernel, 'Periodic': PeriodicKernel}
    if isinstance(kernel, str):
        try:
            kernel = kernel_book[kernel]
        except KeyError:
            print('Select one of the currently available kernels:', *kernel_book.keys())
            raise
    return kernel


def kernel_name(kernel) -> str:
    """'RBF' / 'Matern' for a registry name or one of this module's callables; raises for
    anything else (arbitrary user kernels cannot run in the fused HIP pipeline)."""
    if isinstance(kernel, str):
        if kernel in _lib.KERNEL_KINDS:
            return kernel
        raise NotImplementedError(f"kernel {kernel!r} has no MI355X path; available: {sorted(_lib.KERNEL_KINDS)}")
    name = getattr(kernel, "gpx_name", None)
    if name is None:
        raise NotImplementedError("custom kernel callables cannot run in the fused HIP exact-GP pipeline; "
                                  "use 'RBF' or 'Matern'")
    return name
