"""The example workflows (README / notebook flows of the reference) run end to end on the GPU."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_exactgp_1d_example():
    import exactgp_1d
    out = exactgp_1d.main(num_warmup=150, num_samples=150, verbose=False)
    assert out["rmse"] < 0.1 and out["sd_out"] > out["sd_in"]
    assert out["y_pred"].shape == (200,) and out["y_sampled"].shape == (150, 10, 200) and out["acq"].shape == (200,)


def test_sparse_image_example():
    import sparse_image
    out = sparse_image.main(size=64, keep=0.2, num_steps=120, verbose=False)
    assert out["rmse"] < 0.25 * out["image_sd"]
    assert out["recon"].shape == (64, 64) and np.all(out["var"] > 0)


def test_custom_priors_and_node_sweep_example():
    import custom_priors_node_sweep as ex
    with pytest.warns(UserWarning):  # kernel_prior: the reference's own warning (gp.py:116-123)
        out = ex.main(num_warmup=120, num_samples=120, verbose=False)
    assert out["same"] and out["rmse"] < 0.15 and 1.2 < out["t"] < 2.2


def test_bayesian_optimisation_loop_example():
    import bo_loop
    out = bo_loop.main(num_steps=10, num_warmup=60, num_samples=60, verbose=False)
    assert out["n_measured"] == 14
    assert abs(out["x_best"] - out["x_true"]) <= 0.1 and out["y_best"] <= out["y_true"] + 0.1
