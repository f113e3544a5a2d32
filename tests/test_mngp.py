"""MeasuredNoiseGP / LinReg (gpax/models/mngp.py, linreg.py) — host logic on the test-only OracleEngine,
mirroring gpax/tests/test_mngp.py."""
import numpy as np
import pytest

from gpax_amd import MeasuredNoiseGP, _lib
from gpax_amd.models import LinReg
from gpax_amd.utils import get_keys
from oracle import cpu_ref as ref
from tests.oracle_engine import OracleEngine


@pytest.fixture(autouse=True)
def oracle_engine():
    _lib.set_engine(OracleEngine())
    yield
    _lib.set_engine(None)


def variable_noise(x):
    return 0.1 + 0.5 * x


def get_dummy_data(seed=0):
    rng = np.random.default_rng(seed)
    f = lambda x: np.sin(x) * x
    X = np.linspace(1, 2, 8)
    y_all = np.array([f(x) + rng.normal(0, variable_noise(x), 10) for x in X])
    return X, y_all.mean(1), y_all.var(1)


def test_fit():  # test_mngp.py:28-33
    X, y, measured_noise = get_dummy_data()
    m = MeasuredNoiseGP(1, "RBF")
    m.fit(get_keys()[0], X, y, measured_noise, num_warmup=10, num_samples=10, progress_bar=False, print_summary=False)
    assert m.mcmc is not None
    s = m.get_samples()
    assert set(s) == {"k_length", "k_scale", "noise"}
    assert np.all(s["noise"] == 0.0) and s["noise"].shape == (10,)  # numpyro.deterministic("noise", 0)
    with pytest.raises(ValueError):
        m.fit(get_keys()[0], X, y, measured_noise[:-1], num_warmup=2, num_samples=2, progress_bar=False)


def test_log_joint_uses_the_measured_noise():
    X, y, measured_noise = get_dummy_data()
    m = MeasuredNoiseGP(1, "Matern")
    m.X_train, m.y_train = m._set_data(X, y)
    m.measured_noise = measured_noise
    m._use_measured = True
    sites = m._sites()
    assert [s.name for s in sites] == ["k_length", "k_scale"]
    u = np.array([0.2, -0.1])
    val, grad = m._log_joint(sites, u, 1e-6, jacobian=True)
    theta = m._unpack(sites, u)
    p = {"k_length": theta["k_length"], "k_scale": theta["k_scale"], "noise": 0.0}
    expect = ref.exactgp_log_likelihood(m.X_train, y, p, kernel="Matern", jitter=1e-6, measured_noise=measured_noise)
    off = 0
    for s in sites:
        ui = u[off:off + s.size]
        expect += np.sum(s.dist.log_prob(s.dist.transform(ui))) + np.sum(s.dist.log_abs_det_jacobian(ui)[0])
        off += s.size
    assert abs(val - expect) < 1e-10 * abs(expect)
    fd = np.empty_like(u)
    for i in range(u.size):
        up, um = u.copy(), u.copy()
        up[i] += 1e-6
        um[i] -= 1e-6
        fd[i] = (m._log_joint(sites, up, 1e-6, True)[0] - m._log_joint(sites, um, 1e-6, True)[0]) / 2e-6
    np.testing.assert_allclose(grad, fd, rtol=1e-5)
    # without the measured noise the value differs: the per-point diagonal really is in K
    m._use_measured = False
    assert abs(m._log_joint(sites, u, 1e-6, True)[0] - val) > 1e-3


def test_get_mvn_posterior():  # test_mngp.py:36-53
    X, y, measured_noise = get_dummy_data()
    X_test = get_dummy_data(1)[0]
    params = {"k_length": np.array([1.0]), "k_scale": np.array(1.0), "noise": np.array(0.0)}
    m = MeasuredNoiseGP(1, "RBF")
    m.X_train, m.y_train, m.measured_noise = X[:, None], y, measured_noise
    mean, cov = m.get_mvn_posterior(X_test[:, None], params)
    assert mean.shape == (8,) and cov.shape == (8, 8)
    # the reference does not override get_mvn_posterior: the training block is kernel + jitter I (noise = 0)
    m_ref, c_ref = ref.get_mvn_posterior(X[:, None], y, X_test[:, None], {"k_length": 1.0, "k_scale": 1.0, "noise": 0.0},
                                         False, kernel="RBF", jitter=1e-6, route="chol")
    np.testing.assert_allclose(mean, m_ref, rtol=1e-6)


@pytest.mark.parametrize("n", [1, 5])
def test_predict_single_sample(n):  # test_mngp.py:56-76
    X, y, measured_noise = get_dummy_data()
    X_test = np.linspace(1, 2, 6)
    params = {"k_length": np.array([1.0]), "k_scale": np.array(1.0), "noise": np.array(0.0)}
    m = MeasuredNoiseGP(1, "Matern")
    m.X_train, m.y_train, m.measured_noise = X[:, None], y, measured_noise
    noise_pred = variable_noise(X_test) ** 2
    y_mean, y_sampled = m._predict(get_keys()[1], X_test[:, None], params, noise_pred, n)
    assert y_mean.shape == (6,) and y_sampled.shape == (n, 6)
    eps = np.random.default_rng(0).standard_normal((n, 6))
    m_ref, _ = ref.measured_noise_predict_one(X[:, None], y, X_test[:, None], {"k_length": 1.0, "k_scale": 1.0},
                                              noise_pred, eps, noiseless=False, kernel="Matern")
    np.testing.assert_allclose(y_mean, m_ref, rtol=1e-6)


@pytest.mark.parametrize("noise_pred_fn", ["linreg", "gpreg"])
def test_predict(noise_pred_fn):  # test_mngp.py:79-100
    X, y, measured_noise = get_dummy_data()
    X_test = np.linspace(1, 2, 6)
    rng = np.random.default_rng(0)
    samples = {"k_length": np.exp(0.2 * rng.standard_normal((12, 1))), "k_scale": np.exp(0.2 * rng.standard_normal(12)),
               "noise": np.zeros(12)}
    m = MeasuredNoiseGP(1, "Matern")
    m.X_train, m.y_train, m.measured_noise = X[:, None], y, measured_noise
    kw = dict(num_iterations=300) if noise_pred_fn == "linreg" else dict(num_steps=60)
    y_mean, y_sampled = m.predict(get_keys()[1], X_test, samples, n=3, noise_prediction_method=noise_pred_fn, **kw)
    assert y_mean.shape == (6,) and y_sampled.shape == (12, 3, 6)
    assert m.noise_predicted.shape == (6,)
    # every sample's marginal draw: mean_s + sqrt(var_s + noise_predicted) * eps, with var_s from the oracle posterior
    s = 5
    p = {"k_length": samples["k_length"][s], "k_scale": samples["k_scale"][s], "noise": 0.0}
    m_ref, c_ref = ref.get_mvn_posterior(X[:, None], y, X_test[:, None], p, True, kernel="Matern", jitter=1e-6,
                                         route="chol")
    sig = np.sqrt(np.clip(np.diag(c_ref) + m.noise_predicted, 0, None))
    z = (y_sampled[s] - m_ref[None, :]) / sig[None, :]
    assert np.all(np.isfinite(z)) and np.abs(z).max() < 6.0
    with pytest.raises(NotImplementedError):
        m.predict(get_keys()[1], X_test, samples, noise_prediction_method="spline")


def test_linreg_recovers_a_line():  # gpax/models/linreg.py:18-56
    rng = np.random.default_rng(1)
    x = np.linspace(0, 3, 40)[:, None]
    y = 0.7 + 1.9 * x[:, 0] + 0.05 * rng.standard_normal(40)
    lr = LinReg()
    lr.train(x, y)  # defaults: Adam(0.01), 5000 steps
    p = lr.get_params()
    assert abs(p["alpha"] - 0.7) < 0.1 and abs(p["beta"][0] - 1.9) < 0.1 and 0.01 < p["sigma"] < 0.3
    np.testing.assert_allclose(lr.predict(np.array([[1.0], [2.0]])), [2.6, 4.5], atol=0.15)


def test_second_fit_on_the_same_X_uploads_the_new_measured_noise():
    """the diagonal cache was keyed on object identities, so fit(noise=a); fit(noise=b) on the same X
    array left `a` on the device.  The key now carries a per-fit counter."""
    X, y, _ = get_dummy_data()
    X2 = np.ascontiguousarray(X[:, None], dtype=np.float64)  # _set_data returns this very object
    eng = _lib.get_engine()
    seen = []
    real = eng.set_diag
    eng.set_diag = lambda v: (seen.append(None if v is None else np.array(v)), real(v))[1]
    m = MeasuredNoiseGP(1, "RBF")
    kw = dict(num_warmup=3, num_samples=3, progress_bar=False, print_summary=False)
    m.fit(get_keys()[0], X2, y, np.full(8, 0.01), **kw)
    m.fit(get_keys()[0], X2, y, np.full(8, 5.0), **kw)
    vals = [v[0] for v in seen if v is not None]
    assert 0.01 in vals and 5.0 in vals
    assert vals.index(5.0) > vals.index(0.01)


def test_model_returns_the_log_joint_with_the_measured_noise_on_the_diagonal():
    """MeasuredNoiseGP.model(X, y, measured_noise) (mngp.py:74-98): priors + log N(y | 0, k + jitter I + diag(measured_noise));
    no noise site (the deterministic 0).  The inherited ExactGP.model used to fail on the missing site."""
    from gpax_amd.infer import dist
    X, y, mn = get_dummy_data()
    m = MeasuredNoiseGP(1, "Matern")
    params = {"k_length": np.array([0.7]), "k_scale": 1.4}
    lp = dist.LogNormal(0, 1).log_prob(np.array([0.7]))[0] + dist.LogNormal(0, 1).log_prob(np.array([1.4]))[0]
    assert abs(m.model(X, None, params=params) - lp) < 1e-12
    full = m.model(X, y, mn, params=params)
    p = {"k_length": np.array([0.7]), "k_scale": 1.4, "noise": 0.0}
    expect = ref.exactgp_log_likelihood(X[:, None], y, p, kernel="Matern", jitter=1e-6, measured_noise=mn)
    assert abs(full - (lp + expect)) < 1e-9 * abs(expect)
    assert m.model(X, y, 10.0 * mn, params=params) != full and np.isfinite(m.model(X, y, mn))
    with pytest.raises(ValueError):
        m.model(X, y, params=params)
    with pytest.raises(ValueError):
        m.model(X, y, mn[:-1], params=params)
