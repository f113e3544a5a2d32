"""GPU: the other axes of the path that shard over the GPUs of a node (SURVEY.md 8e "later"), each
from one process with one context and one host thread per GPU, no collective:
  * NUTS chains — ExactGP.fit(num_chains = k, chain_method = 'parallel' | device = [...]): chain c on GPU devices[c % G],
    the chains of one GPU in lockstep as one batched device pass (gpax/models/gp.py:173-174,214: NumPyro pmaps them);
  * viGP.predict_in_batches(device = 'all' | [...]): contiguous blocks of the slices of X_new per GPU, K(theta) factored
    on every GPU (gpax/models/vigp.py:129-151 re-inverts per slice).
On the 1-GPU test box the device lists name GPU 0 several times (separate contexts: the threading / grouping logic);
with more GPUs visible, all of them."""
import numpy as np
import pytest

from bench_inputs import synthetic_problem

pytestmark = pytest.mark.gpu


def _devices():
    from gpax_amd import _lib
    n = _lib.visible_device_count()
    return list(range(n)) if n > 1 else [0, 0, 0]


def test_chains_dealt_over_devices_draw_what_sequential_chains_draw():
    from gpax_amd import ExactGP, _lib
    _lib.set_engine(None)
    X, y, _, _ = synthetic_problem(300, 2, 4, seed=2)
    out = []
    for kw in (dict(chain_method="sequential"), dict(chain_method="vectorized"),
               dict(chain_method="parallel", device=_devices())):
        m = ExactGP(2, "Matern")
        m.fit(7, X, y, num_warmup=12, num_samples=10, num_chains=4, progress_bar=False, print_summary=False, **kw)
        out.append(m.get_samples(chain_dim=True))
    for other in out[1:]:
        for k in out[0]:
            np.testing.assert_array_equal(out[0][k], other[k])
    assert out[0]["k_length"].shape == (4, 10, 2)


def test_measured_noise_chains_over_devices_keep_their_diagonal():
    from gpax_amd import _lib
    from gpax_amd.models.mngp import MeasuredNoiseGP
    _lib.set_engine(None)
    X, y, _, _ = synthetic_problem(200, 1, 4, seed=4)
    noise = 0.05 + 0.1 * np.random.default_rng(0).uniform(size=200)
    out = []
    for kw in (dict(chain_method="sequential"), dict(chain_method="parallel", device=_devices()[:2])):
        m = MeasuredNoiseGP(1, "RBF")
        m.fit(3, X, y, noise, num_warmup=10, num_samples=8, num_chains=2, progress_bar=False, print_summary=False, **kw)
        out.append(m.get_samples(chain_dim=True))
    for k in out[0]:
        np.testing.assert_array_equal(out[0][k], out[1][k])


def test_vigp_predict_in_batches_over_devices_equals_one_gpu():
    from gpax_amd import _lib, viGP
    _lib.set_engine(None)
    X, y, _, p = synthetic_problem(1500, 2, 4, seed=6)
    Xn = np.random.default_rng(1).uniform(0, 10, (2350, 2))  # 24 slices of 100, the last one ragged
    m = viGP(2, "Matern", mean_fn=lambda x: 0.1 * x[:, 0])
    m.X_train, m.y_train = m._set_data(X, y)
    m._data_version += 1
    theta = {"k_length": p["k_length"], "k_scale": np.float64(p["k_scale"]), "noise": np.float64(p["noise"])}
    a = m.predict_in_batches(0, Xn, batch_size=100, samples=theta)
    b = m.predict_in_batches(0, Xn, batch_size=100, samples=theta, device=_devices())
    c = m.predict_in_batches(0, Xn, batch_size=100, samples=theta, device="all")
    for u, v, w in zip(a, b, c):
        np.testing.assert_array_equal(u, v)
        np.testing.assert_array_equal(u, w)
    bad = dict(theta, k_scale=np.float64(-1.0))  # not positive definite on every GPU: NaN, not a crash
    mean, var = m.predict_in_batches(0, Xn[:300], batch_size=100, samples=bad, device=_devices())
    assert np.isnan(mean).all() and np.isnan(var).all()
