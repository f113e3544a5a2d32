"""viGP through its public surface on the checker-backed engine (no GPU): the behaviours gpax's own viGP tests assert
(gpax/tests/test_vigp.py — fit with array / 2-D inputs and every kernel, the point estimates, predict shapes with and
without noise, slices, deterministic and probabilistic mean functions, prior draws, jitter sensitivity, guide types), each
with the reference test it follows cited.  Values against the oracle are in tests/test_host_logic.py / test_gpu_models.py."""
import numpy as np
import pytest

import gpax_amd
from gpax_amd import _lib, dist
from gpax_amd.models import viGP
from gpax_amd.utils import get_keys
from tests.oracle_engine import OracleEngine


@pytest.fixture(autouse=True)
def oracle_engine():
    _lib.set_engine(OracleEngine())
    yield
    _lib.set_engine(None)


def data(unsqueeze=False, as_list=False, seed=0):
    rng = np.random.default_rng(seed)
    X = np.linspace(1, 2, 8) + 0.1 * rng.standard_normal(8)
    y = 10 * X ** 2
    if unsqueeze:
        X = X[:, None]
    return (X.tolist(), y.tolist()) if as_list else (X, y)


def power_mean(x, params):
    return params["a"] * x[:, 0] ** params["b"]


def power_mean_priors():
    return {"a": gpax_amd.sample("a", dist.LogNormal(0, 1)), "b": gpax_amd.sample("b", dist.Normal(3, 1))}


FIT = dict(num_steps=30, progress_bar=False, print_summary=False)


@pytest.mark.parametrize("as_list", [True, False])  # the reference feeds jax and NumPy arrays; here: anything np.asarray takes
@pytest.mark.parametrize("unsqueeze", [True, False])
@pytest.mark.parametrize("kernel", ["RBF", "Matern", "Periodic"])
def test_fit_and_point_estimates(kernel, as_list, unsqueeze):  # test_vigp.py:43-66
    if as_list and unsqueeze:
        pytest.skip("same input as the array case")
    X, y = data(unsqueeze, as_list)
    m = viGP(1, kernel)
    m.fit(get_keys()[0], X, y, **FIT)
    assert m.svi is not None
    s = m.get_samples()
    assert isinstance(s, dict)
    for k, v in s.items():
        assert isinstance(v, np.ndarray), k
    assert ("period" in s) == (kernel == "Periodic")


@pytest.mark.parametrize("unsqueeze", [True, False])
@pytest.mark.parametrize("noiseless", [False, True])
def test_prediction_from_given_parameters(unsqueeze, noiseless):  # test_vigp.py:68-100
    X, y = data(unsqueeze)
    Xt, _ = data(unsqueeze, seed=1)
    params = {"k_length": np.array([[1.0]]), "k_scale": np.array([1.0]), "noise": np.array([0.1])}
    m = viGP(1, "RBF")
    m.X_train, m.y_train = m._set_data(X, y)
    mean, var = m.predict(get_keys()[1], Xt, params, noiseless=noiseless)
    assert isinstance(mean, np.ndarray) and isinstance(var, np.ndarray)
    assert mean.shape == np.squeeze(Xt).shape and var.shape == np.squeeze(Xt).shape
    if noiseless:  # the noisy variance carries the noise on top (vigp.py:153-185 through gp.py:261)
        _, var_noisy = m.predict(get_keys()[1], Xt, params, noiseless=False)
        np.testing.assert_allclose(var_noisy - var, 0.1, rtol=1e-8)


@pytest.mark.parametrize("unsqueeze", [True, False])
@pytest.mark.parametrize("batch_size", [2, 3, 8])
def test_prediction_in_batches(unsqueeze, batch_size):  # test_vigp.py:102-119
    X, y = data(unsqueeze)
    Xt, _ = data(unsqueeze, seed=1)
    params = {"k_length": np.array([1.0]), "k_scale": np.array(1.0), "noise": np.array(0.1)}
    m = viGP(1, "RBF")
    m.X_train, m.y_train = m._set_data(X, y)
    mean, var = m.predict_in_batches(get_keys()[1], Xt, batch_size, params)
    whole = m.predict(get_keys()[1], Xt, params)
    assert mean.shape == np.squeeze(Xt).shape and var.shape == np.squeeze(Xt).shape
    np.testing.assert_allclose(mean, whole[0], rtol=1e-10)
    np.testing.assert_allclose(var, whole[1], rtol=1e-10)


@pytest.mark.parametrize("kernel", ["RBF", "Matern", "Periodic"])
def test_fit_predict(kernel):  # test_vigp.py:121-133
    X, y = data()
    Xt, _ = data(seed=1)
    m = viGP(1, kernel)
    m.fit(get_keys()[0], X, y, **FIT)
    mean, var = m.predict(get_keys()[1], Xt)
    assert mean.shape == Xt.shape and var.shape == Xt.shape and np.all(np.isfinite(mean)) and np.all(var > 0)


@pytest.mark.parametrize("batch_size", [2, 3, 8])
def test_fit_predict_in_batches(batch_size):  # test_vigp.py:135-148
    X, y = data()
    Xt, _ = data(seed=1)
    m = viGP(1, "RBF")
    m.fit(get_keys()[0], X, y, **FIT)
    mean, var = m.predict_in_batches(get_keys()[1], Xt, batch_size=batch_size)
    assert mean.shape == Xt.shape and var.shape == Xt.shape


@pytest.mark.parametrize("probabilistic", [False, True])
def test_fit_predict_with_mean_functions(probabilistic):  # test_vigp.py:151-193
    X, y = data()
    Xt, _ = data(seed=1)
    if probabilistic:
        m = viGP(1, "RBF", mean_fn=power_mean, mean_fn_prior=power_mean_priors)
    else:
        m = viGP(1, "RBF", mean_fn=lambda x: 8 * x[:, 0] ** 2)
    m.fit(get_keys()[0], X, y, **FIT)
    assert m.svi is not None
    if probabilistic:
        assert {"a", "b"} <= set(m.get_samples())
    mean, var = m.predict(get_keys()[1], Xt)
    assert mean.shape == Xt.shape and var.shape == Xt.shape


def test_sample_from_prior():  # test_vigp.py:196-201
    X, _ = data()
    prior_pred = viGP(1, "RBF").sample_from_prior(get_keys()[0], X, num_samples=8)
    assert prior_pred.shape == (8, X.shape[0])


def test_jitter_changes_the_fit_and_the_prediction():  # test_vigp.py:204-233
    X, y = data()
    fits = []
    for jitter in (1e-6, 1e-6, 1e-4):
        m = viGP(1, "RBF")
        m.fit(get_keys()[0], X, y, jitter=jitter, **FIT)
        fits.append(m.get_samples()["k_length"])
    assert np.all(fits[0] - fits[1] == 0) and np.all(fits[0] - fits[2] != 0)
    Xu, yu = data(unsqueeze=True)
    samples = {"k_length": np.array([[0.8]]), "k_scale": np.array([1.3]), "noise": np.array([0.2])}
    m = viGP(1, "RBF")
    m.X_train, m.y_train = m._set_data(Xu, yu)
    mean1, var1 = m.predict(get_keys()[1], Xu, samples, jitter=1e-6)
    mean2, var2 = m.predict(get_keys()[1], Xu, samples, jitter=1e-4)
    assert np.count_nonzero(var1 - var2) > 0 and np.count_nonzero(mean1 - mean2) > 0


def test_guide_type():  # test_vigp.py:236-250
    X, y = data()
    fits = []
    for guide in ("delta", "delta", "normal"):
        m = viGP(1, "RBF", guide=guide)
        m.fit(get_keys()[0], X, y, **FIT)
        fits.append(m.get_samples()["k_length"])
    assert np.all(fits[0] - fits[1] == 0) and np.all(fits[0] - fits[2] != 0)
    assert viGP(1, "RBF", guide="laplace").guide_type == "delta"  # anything but 'normal' is the MAP guide (vigp.py:74)
