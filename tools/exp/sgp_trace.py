"""Experiment: 3 viSparseGP steps (bound + gradient) at C5 size under rocprofv3 --kernel-trace."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from bench_inputs import synthetic_sparse_image
from gpax_amd import _lib
from gpax_amd.utils import get_keys, initialize_inducing_points, preprocess_sparse_image
img, sparse = synthetic_sparse_image(512, 512, 0.0625, seed=3)
X, y, X_full = preprocess_sparse_image(sparse)
y = y - y.mean()
Xu = initialize_inducing_points(X, 0.125, "random", get_keys(0)[0])
eng = _lib.get_engine(0)
eng.set_train(X)
for i in range(3):
    eng.sgp_bound(1, [25.0, 25.0], 1.0, 1e-2 * (1 + 1e-13 * i), 1e-6, Xu, y, True)  # (bit-identical inputs would reuse the forward pass)
