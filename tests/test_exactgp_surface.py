"""ExactGP through its public surface on the checker-backed engine (no GPU): the behaviours gpax's own ExactGP tests assert
(gpax/tests/test_gp.py), each with the reference test it follows cited — fit on 1-D / 2-D inputs with every kernel, the
sample dictionary with and without the chain axis, the MVN posterior (shape, determinism, the noiseless variant), the
single-sample predictor, predict / predict_in_batches over 100 samples drawn at random (negative "variances" included:
NaN rows, never a crash), mean functions, prior draws, jitter sensitivity.  Values against the oracle live in
tests/test_host_logic.py and the -m gpu tests."""
import numpy as np
import pytest

import gpax_amd
from gpax_amd import _lib, dist
from gpax_amd.models import ExactGP
from gpax_amd.utils import get_keys
from tests.oracle_engine import OracleEngine


@pytest.fixture(autouse=True)
def oracle_engine():
    _lib.set_engine(OracleEngine())
    yield
    _lib.set_engine(None)


def data(unsqueeze=False, seed=0):
    rng = np.random.default_rng(seed)
    X = np.linspace(1, 2, 8) + 0.1 * rng.standard_normal(8)
    return (X[:, None] if unsqueeze else X), 10 * X ** 2


def power_mean(x, params):
    return params["a"] * x[:, 0] ** params["b"]


def power_mean_priors():
    return {"a": gpax_amd.sample("a", dist.LogNormal(0, 1)), "b": gpax_amd.sample("b", dist.Normal(3, 1))}


ONE = {"k_length": np.array([1.0]), "k_scale": np.array(1.0), "noise": np.array(0.1)}
NUTS = dict(num_warmup=15, num_samples=15, progress_bar=False, print_summary=False)


def random_samples(S=100, seed=3):
    """What the reference's tests feed predict(): 100 standard-normal 'samples' — half of the variances negative."""
    rng = np.random.default_rng(seed)
    return {"k_length": rng.standard_normal((S, 1)), "k_scale": rng.standard_normal(S), "noise": rng.standard_normal(S)}


def model_with_data(unsqueeze=True):
    X, y = data(unsqueeze)
    m = ExactGP(1, "RBF")
    m.X_train, m.y_train = m._set_data(X, y)
    return m


@pytest.mark.parametrize("unsqueeze", [True, False])
@pytest.mark.parametrize("kernel", ["RBF", "Matern", "Periodic"])
def test_fit_and_get_samples(kernel, unsqueeze):  # test_gp.py:41-64
    X, y = data(unsqueeze)
    m = ExactGP(1, kernel)
    m.fit(get_keys()[0], X, y, **NUTS)
    assert m.mcmc is not None
    samples = m.get_samples()
    assert isinstance(samples, dict)
    for k, v in samples.items():
        assert isinstance(v, np.ndarray) and v.shape[0] == 15, k


@pytest.mark.parametrize("chain_dim, samples_dim", [(True, 2), (False, 1)])
def test_get_samples_chain_dim(chain_dim, samples_dim):  # test_gp.py:67-76
    X, y = data()
    m = ExactGP(1, "RBF")
    m.fit(get_keys()[0], X, y, num_warmup=8, num_samples=8, num_chains=2, progress_bar=False, print_summary=False)
    s = m.get_samples(chain_dim)
    assert s["k_scale"].ndim == samples_dim and s["noise"].ndim == samples_dim and s["k_length"].ndim == samples_dim + 1


def test_get_mvn_posterior_and_its_noiseless_variant():  # test_gp.py:139-170
    m = model_with_data()
    Xt, _ = data(unsqueeze=True, seed=1)
    mean, cov = m.get_mvn_posterior(Xt, ONE)
    assert isinstance(mean, np.ndarray) and isinstance(cov, np.ndarray)
    assert mean.shape == (Xt.shape[0],) and cov.shape == (Xt.shape[0], Xt.shape[0])
    mean_, cov_ = m.get_mvn_posterior(Xt, ONE, noiseless=False)
    mean2, cov2 = m.get_mvn_posterior(Xt, ONE, noiseless=True)
    np.testing.assert_array_equal(mean, mean_)
    np.testing.assert_array_equal(cov, cov_)
    np.testing.assert_array_equal(mean, mean2)
    assert np.count_nonzero(cov - cov2) > 0


def test_single_sample_prediction():  # test_gp.py:173-187
    m = model_with_data()
    Xt, _ = data(unsqueeze=True, seed=1)
    y_mean, y_sample = m._predict(get_keys()[0], Xt, ONE, 1)
    assert isinstance(y_mean, np.ndarray) and isinstance(y_sample, np.ndarray)
    assert y_mean.shape == (Xt.shape[0],) and y_sample.shape == (1, Xt.shape[0])


@pytest.mark.parametrize("n", [1, 10])
@pytest.mark.parametrize("unsqueeze", [True, False])
def test_prediction(unsqueeze, n):  # test_gp.py:190-206
    m = model_with_data()
    Xt, _ = data(unsqueeze=unsqueeze, seed=1)
    y_mean, y_sampled = m.predict(get_keys()[1], Xt, random_samples(), n=n)
    assert isinstance(y_mean, np.ndarray) and isinstance(y_sampled, np.ndarray)
    assert y_mean.shape == np.squeeze(Xt).shape and y_sampled.shape == (100, n, Xt.shape[0])


def test_noiseless_prediction():  # test_gp.py:209-222
    m = model_with_data()
    Xt, _ = data(unsqueeze=True, seed=1)
    rng = np.random.default_rng(5)  # positive samples: the comparison is about the noise term, not about NaN rows
    s = {"k_length": rng.uniform(0.5, 1.5, (100, 1)), "k_scale": rng.uniform(0.5, 1.5, 100), "noise": rng.uniform(0.05, 0.3, 100)}
    y_mean1, y_sampled1 = m.predict(get_keys()[1], Xt, s, n=1, noiseless=True)
    y_mean2, y_sampled2 = m.predict(get_keys()[1], Xt, s, n=1, noiseless=False)
    np.testing.assert_array_equal(y_mean1, y_mean2)
    assert np.count_nonzero(y_sampled1 - y_sampled2) > 0


@pytest.mark.parametrize("batch_size", [2, 3, 8])
@pytest.mark.parametrize("n", [1, 10])
def test_prediction_in_batches(batch_size, n):  # test_gp.py:225-241
    m = model_with_data()
    Xt, _ = data(seed=1)
    y_pred, y_sampled = m.predict_in_batches(get_keys()[1], Xt, batch_size, random_samples(), n=n)
    assert y_pred.shape == Xt.shape and y_sampled.shape == (100, n, Xt.shape[0])


@pytest.mark.parametrize("kernel", ["RBF", "Matern", "Periodic"])
def test_fit_predict(kernel):  # test_gp.py:244-255
    X, y = data()
    Xt, _ = data(seed=1)
    m = ExactGP(1, kernel)
    m.fit(get_keys()[0], X, y, **NUTS)
    y_pred, y_sampled = m.predict(get_keys()[1], Xt)
    assert y_pred.shape == Xt.shape and y_sampled.shape == (15, 1, Xt.shape[0])


@pytest.mark.parametrize("n", [1, 10])
@pytest.mark.parametrize("noiseless", [False, True])
def test_fit_predict_in_batches(n, noiseless):  # test_gp.py:258-282
    X, y = data()
    Xt, _ = data(seed=1)
    m = ExactGP(1, "RBF")
    m.fit(get_keys()[0], X, y, **NUTS)
    y_pred, y_sampled = m.predict_in_batches(get_keys()[1], Xt, batch_size=4, n=n, noiseless=noiseless)
    assert y_pred.shape == Xt.shape and y_sampled.shape == (15, n, Xt.shape[0])


@pytest.mark.parametrize("probabilistic", [False, True])
def test_fit_predict_with_mean_functions(probabilistic):  # test_gp.py:285-326
    X, y = data()
    Xt, _ = data(seed=1)
    if probabilistic:
        m = ExactGP(1, "RBF", mean_fn=power_mean, mean_fn_prior=power_mean_priors)
    else:
        m = ExactGP(1, "RBF", mean_fn=lambda x: 8 * x[:, 0] ** 2)
    m.fit(get_keys()[0], X, y, **NUTS)
    assert m.mcmc is not None
    if probabilistic:
        assert {"a", "b"} <= set(m.get_samples())
    y_pred, y_sampled = m.predict(get_keys()[1], Xt)
    assert y_pred.shape == Xt.shape and y_sampled.shape == (15, 1, Xt.shape[0])


def test_sample_from_prior():  # test_gp.py:329-334
    X, _ = data()
    prior_pred = ExactGP(1, "RBF").sample_from_prior(get_keys()[0], X, num_samples=8)
    assert prior_pred.shape == (8, X.shape[0])


def test_jitter_changes_the_fit_and_the_prediction():  # test_gp.py:337-366
    X, y = data()
    fits = []
    for jitter in (1e-6, 1e-6, 1e-4):
        m = ExactGP(1, "RBF")
        m.fit(get_keys()[0], X, y, jitter=jitter, **NUTS)
        fits.append(m.get_samples()["k_length"])
    np.testing.assert_array_equal(fits[0], fits[1])
    assert np.count_nonzero(fits[0] - fits[2]) > 0
    m = model_with_data()
    Xt, _ = data(unsqueeze=True, seed=1)
    s = {"k_length": np.array([[0.8]]), "k_scale": np.array([1.3]), "noise": np.array([0.2])}
    _, y1 = m.predict(get_keys()[1], Xt, s, n=1, jitter=1e-6)
    _, y2 = m.predict(get_keys()[1], Xt, s, n=1, jitter=1e-4)
    assert np.count_nonzero(y1 - y2) > 0
