"""VarNoiseGP (gpax/models/hskgp.py) — host logic on the test-only OracleEngine, mirroring
gpax/tests/test_hskgp.py."""
import numpy as np
import pytest

from gpax_amd import VarNoiseGP, _lib
from gpax_amd.infer import dist
from gpax_amd.utils import get_keys
from oracle import cpu_ref as ref
from tests.oracle_engine import OracleEngine


@pytest.fixture(autouse=True)
def oracle_engine():
    _lib.set_engine(OracleEngine())
    yield
    _lib.set_engine(None)


def get_dummy_data(unsqueeze=False, seed=0):
    rng = np.random.default_rng(seed)
    X = np.linspace(1, 2, 8) + 0.1 * rng.standard_normal(8)
    y = 10 * X ** 2
    return (X[:, None] if unsqueeze else X), y


def noise_fn(x, params):
    return np.exp(params["a"] + params["b"] * x[:, 0])  # positive: its log is the prior mean of log_var


noise_fn_prior = {"a": dist.Normal(0.0, 1.0), "b": dist.Normal(0.0, 1.0)}


@pytest.mark.parametrize("noise_kernel", ["RBF", "Matern"])
def test_fit(noise_kernel):  # test_hskgp.py:34-40
    X, y = get_dummy_data()
    m = VarNoiseGP(1, "RBF", noise_kernel=noise_kernel)
    m.fit(get_keys()[0], X, y, num_warmup=10, num_samples=10, progress_bar=False, print_summary=True)
    assert m.mcmc is not None
    s = m.get_samples()
    assert set(s) == {"k_noise_scale", "k_noise_length", "log_var", "k_length", "k_scale"}
    assert s["log_var"].shape == (10, 8)


def test_fit_with_custom_noise_lscale_and_mean_fns():  # test_hskgp.py:43-65
    X, y = get_dummy_data()
    m = VarNoiseGP(1, "RBF", noise_kernel="Matern", noise_lengthscale_prior_dist=dist.HalfNormal(1.0),
                   noise_mean_fn=noise_fn, noise_mean_fn_prior=noise_fn_prior,
                   mean_fn=lambda x, p: p["c"] * x[:, 0] ** 2, mean_fn_prior={"c": dist.Normal(10.0, 1.0)})
    m.fit(get_keys()[0], X, y, num_warmup=10, num_samples=10, progress_bar=False, print_summary=False)
    s = m.get_samples()
    assert {"a", "b", "c"} <= set(s)
    v = m.get_data_var_samples()  # test_hskgp.py:133-150
    assert v.shape == (10, 8) and np.all(v > 0)


def test_log_joint_matches_the_two_mvn_terms_and_gradient_matches_fd():
    # well-separated inputs and short lengthscales: cond(K) small enough for central differences to resolve 1e-5
    X = np.linspace(0.0, 7.0, 8)
    y = np.sin(X) + 0.1 * np.cos(5 * X)
    m = VarNoiseGP(1, "Matern", noise_kernel="RBF", noise_mean_fn=noise_fn, noise_mean_fn_prior=noise_fn_prior)
    m.X_train, m.y_train = m._set_data(X, y)
    sites = m._sites()
    rng = np.random.default_rng(3)
    u = 0.3 * rng.standard_normal(sum(s.size for s in sites))
    u[1] = np.log(0.6)  # k_noise_length
    val, grad = m._log_joint(sites, u, 1e-6, jacobian=True)
    theta = m._unpack(sites, u)
    p = {k: theta[k] for k in ("k_length", "k_scale", "k_noise_length", "k_noise_scale", "log_var")}
    expect = ref.varnoise_log_likelihood(m.X_train, y, p, kernel="Matern", noise_kernel_name="RBF", jitter=1e-6,
                                         noise_loc=np.log(noise_fn(m.X_train, theta)))
    off = 0
    for s in sites:
        ui = u[off:off + s.size]
        expect += np.sum(s.dist.log_prob(s.dist.transform(ui))) + np.sum(s.dist.log_abs_det_jacobian(ui)[0])
        off += s.size
    assert abs(val - expect) < 1e-9 * abs(expect)
    fd = np.empty_like(u)
    for i in range(u.size):
        up, um = u.copy(), u.copy()
        up[i] += 1e-6
        um[i] -= 1e-6
        fd[i] = (m._log_joint(sites, up, 1e-6, True)[0] - m._log_joint(sites, um, 1e-6, True)[0]) / 2e-6
    np.testing.assert_allclose(grad, fd, rtol=5e-5, atol=5e-5 * np.abs(fd).max())


def test_get_mvn_posterior():  # test_hskgp.py:68-130
    X, y = get_dummy_data(unsqueeze=True)
    X_test, _ = get_dummy_data(unsqueeze=True, seed=1)
    params = {"k_length": np.array([1.0]), "k_scale": np.array(1.0), "noise": np.array(0.1),
              "k_noise_length": np.array(0.5), "k_noise_scale": np.array(1.0), "log_var": np.ones(len(X))}
    m = VarNoiseGP(1, "RBF", noise_kernel="RBF")
    m.X_train, m.y_train = X, y
    mean, cov = m.get_mvn_posterior(X_test, params)
    assert mean.shape == (8,) and cov.shape == (8, 8)
    p = {"k_length": 1.0, "k_scale": 1.0, "k_noise_length": 0.5, "k_noise_scale": 1.0, "log_var": np.ones(8)}
    m_ref, c_ref = ref.varnoise_get_mvn_posterior(X, y, X_test, p, kernel="RBF", noise_kernel_name="RBF")
    # the training block is kernel + 1e-6 I (cond ~ 1e7): inverse vs Cholesky routes agree to ~1e-5
    np.testing.assert_allclose(mean, m_ref, rtol=1e-4)
    np.testing.assert_allclose(np.diag(cov), np.diag(c_ref), rtol=1e-4)
    # with mean functions on both GPs
    m2 = VarNoiseGP(1, "RBF", noise_kernel="RBF", noise_mean_fn=noise_fn, noise_mean_fn_prior=noise_fn_prior,
                    mean_fn=lambda x, p_: p_["c"] * x[:, 0] ** 2, mean_fn_prior={"c": dist.Normal(10.0, 1.0)})
    m2.X_train, m2.y_train = X, y
    params2 = dict(params, a=np.array(0.1), b=np.array(0.2), c=np.array(9.5))
    mean2, cov2 = m2.get_mvn_posterior(X_test, params2)
    assert mean2.shape == (8,) and cov2.shape == (8, 8) and np.all(np.isfinite(cov2))


@pytest.mark.parametrize("n", [1, 4])
def test_predict_shapes_and_consistency_with_single_sample_path(n):
    X, y = get_dummy_data()
    X_test = np.linspace(1.05, 1.95, 6)
    m = VarNoiseGP(1, "Matern", noise_kernel="RBF")
    m.fit(get_keys()[0], X, y, num_warmup=10, num_samples=10, progress_bar=False, print_summary=False)
    y_mean, y_sampled = m.predict(get_keys()[1], X_test, n=n)
    assert y_mean.shape == (6,) and y_sampled.shape == (10, n, 6)
    s = m.get_samples()
    means = []
    for i in range(10):
        p = {k: v[i] for k, v in s.items()}
        mi, Ki = m.get_mvn_posterior(X_test, p)
        means.append(mi)
        z = (y_sampled[i] - mi[None, :]) / np.sqrt(np.diag(Ki))[None, :]
        assert np.all(np.isfinite(z)) and np.abs(z).max() < 6.0
    np.testing.assert_allclose(y_mean, np.mean(means, axis=0), rtol=1e-6, atol=1e-8)
    ym1, ys1 = m._predict(get_keys()[1], X_test, {k: v[0] for k, v in s.items()}, n)
    assert ym1.shape == (6,) and ys1.shape == (n, 6)


def test_unsupported_arguments():
    with pytest.raises(NotImplementedError):
        VarNoiseGP(1, "RBF", noise_kernel="Periodic")
    with pytest.raises(NotImplementedError):
        VarNoiseGP(1, "RBF", noise_kernel_prior=lambda: {})


def test_model_returns_the_log_joint_of_both_gps():
    """VarNoiseGP.model(X, y) (hskgp.py:105-153): priors + log N(log_var | log noise_fn, k_noise) + log N(y | 0, k + diag(exp
    log_var)); y = None drops the last term.  The model's own training data are left alone."""
    X = np.linspace(0.0, 7.0, 8)
    y = np.sin(X) + 0.1 * np.cos(5 * X)
    m = VarNoiseGP(1, "Matern", noise_kernel="RBF")
    m.X_train, m.y_train = m._set_data(X[:5], y[:5])
    lv = 0.2 * np.cos(X)
    params = {"k_length": np.array([0.8]), "k_scale": 1.2, "k_noise_length": 0.6, "k_noise_scale": 0.9, "log_var": lv}
    full = m.model(X, y, params=params)
    expect = ref.varnoise_log_likelihood(X[:, None], y, params, kernel="Matern", noise_kernel_name="RBF", jitter=1e-6,
                                         noise_loc=np.zeros(8))
    lp = sum(dist.LogNormal(0, 1).log_prob(np.array([v]))[0] for v in (0.8, 1.2, 0.6, 0.9))
    assert abs(full - (lp + expect)) < 1e-9 * abs(lp + expect)
    p2 = {"k_length": np.array([0.6]), "k_scale": 0.9, "noise": 0.0}
    lml2 = ref.exactgp_log_likelihood(X[:, None], lv, p2, kernel="RBF", jitter=1e-6)
    assert abs(m.model(X, None, params=params) - (lp + lml2)) < 1e-9 * abs(lp + lml2)
    assert m.X_train.shape == (5, 1) and m.y_train.shape == (5,)  # untouched
    assert np.isfinite(m.model(X, y))
