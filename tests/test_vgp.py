"""vExactGP (gpax/models/vgp.py) — host logic on the test-only OracleEngine, mirroring gpax/tests/test_vgp.py."""
import numpy as np
import pytest

from gpax_amd import _lib, vExactGP
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
    X = np.array([np.linspace(1, 2, 8) + 0.1 * rng.standard_normal(8) for _ in range(3)])
    y = 10 * X ** 2
    return (X[..., None] if unsqueeze else X), y


@pytest.mark.parametrize("unsqueeze", [True, False])
@pytest.mark.parametrize("kernel", ["RBF", "Matern", "Periodic"])
def test_fit_and_get_samples(kernel, unsqueeze):  # test_vgp.py:26-49
    X, y = get_dummy_data(unsqueeze)
    m = vExactGP(1, kernel)
    m.fit(get_keys()[0], X, y, num_warmup=10, num_samples=10, progress_bar=False, print_summary=False)
    assert m.mcmc is not None
    samples = m.get_samples()
    assert set(samples) == {"k_length", "k_scale", "noise"} | ({"period"} if kernel == "Periodic" else set())
    for v in samples.values():
        assert v.shape[:2] == (10, 3)
    assert samples["k_length"].shape == (10, 3, 1)


@pytest.mark.parametrize("chain_dim, samples_dim", [(True, 3), (False, 2)])
def test_get_samples_chain_dim(chain_dim, samples_dim):  # test_vgp.py:52-62
    X, y = get_dummy_data()
    m = vExactGP(1, "RBF")
    m.fit(get_keys()[0], X, y, num_warmup=8, num_samples=8, num_chains=2, progress_bar=False, print_summary=False)
    samples = m.get_samples(chain_dim)
    assert samples["k_scale"].ndim == samples_dim
    assert samples["noise"].ndim == samples_dim
    assert samples["k_length"].ndim == samples_dim + 1


def test_lockstep_chains_equal_sequential():
    X, y = get_dummy_data()
    outs = []
    for method in ["sequential", "vectorized"]:
        m = vExactGP(1, "RBF")
        m.fit(get_keys()[0], X, y, num_warmup=8, num_samples=8, num_chains=2, chain_method=method,
              progress_bar=False, print_summary=False)
        outs.append(m.get_samples(chain_dim=True))
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k])


def test_task_dims_must_match():  # vgp.py:204-206
    m = vExactGP(1, "RBF")
    with pytest.raises(AssertionError):
        m.fit(get_keys()[0], np.zeros((3, 8)), np.zeros((2, 8)), num_warmup=1, num_samples=1)


def test_sites_follow_the_reference_prior_assignment():
    # vgp.py:111-113: k_length ~ LogNormal(0,1) always; `lengthscale_prior_dist` lands on k_scale
    X, y = get_dummy_data()
    m = vExactGP(1, "RBF", lengthscale_prior_dist=dist.HalfNormal(0.3), noise_prior_dist=dist.HalfNormal(0.1))
    m.X_train, m.y_train = m._set_data(X, y)
    sites = {s.name: s for s in m._sites()}
    assert isinstance(sites["k_length"].dist, dist.LogNormal) and sites["k_length"].shape == (3, 1)
    assert isinstance(sites["k_scale"].dist, dist.HalfNormal) and sites["k_scale"].shape == (3,)
    assert isinstance(sites["noise"].dist, dist.HalfNormal) and sites["noise"].shape == (3,)


def test_log_joint_is_the_sum_of_task_likelihoods_and_gradient_matches_fd():
    X, y = get_dummy_data()
    m = vExactGP(1, "Matern")
    m.X_train, m.y_train = m._set_data(X, y)
    sites = m._sites()
    rng = np.random.default_rng(1)
    u = 0.3 * rng.standard_normal(sum(s.size for s in sites))
    val, grad = m._log_joint(sites, u, 1e-6, jacobian=True)
    theta = m._unpack(sites, u)
    expect = 0.0
    for t in range(3):
        p = {"k_length": theta["k_length"][t], "k_scale": theta["k_scale"][t], "noise": theta["noise"][t]}
        expect += ref.exactgp_log_likelihood(m.X_train[t], y[t], p, kernel="Matern", jitter=1e-6)
    off = 0
    for s in sites:
        ui = u[off:off + s.size]
        expect += np.sum(s.dist.log_prob(s.dist.transform(ui))) + np.sum(s.dist.log_abs_det_jacobian(ui)[0])
        off += s.size
    assert abs(val - expect) < 1e-9 * abs(expect)
    fd = np.empty_like(u)
    for i in range(u.size):
        h = 1e-6
        up, um = u.copy(), u.copy()
        up[i] += h
        um[i] -= h
        fd[i] = (m._log_joint(sites, up, 1e-6, True)[0] - m._log_joint(sites, um, 1e-6, True)[0]) / (2 * h)
    np.testing.assert_allclose(grad, fd, rtol=2e-5, atol=2e-5 * np.abs(fd).max())


def test_get_mvn_posterior():  # test_vgp.py:106-136
    X, y = get_dummy_data(unsqueeze=True)
    X_test, _ = get_dummy_data(unsqueeze=True, seed=1)
    params = {"k_length": np.array([[1.0], [1.0], [1.0]]), "k_scale": np.array([1.0, 1.0, 1.0]),
              "noise": np.array([0.1, 0.1, 0.1])}
    m = vExactGP(1, "RBF")
    m.X_train, m.y_train = X, y
    mean, cov = m.get_mvn_posterior(X_test, params)
    assert mean.shape == X_test.shape[:-1]
    assert cov.shape == (X_test.shape[0], X_test.shape[1], X_test.shape[1])
    for t in range(3):
        p = {"k_length": np.array([1.0]), "k_scale": 1.0, "noise": 0.1}
        m_ref, c_ref = ref.get_mvn_posterior(X[t], y[t], X_test[t], p, False, kernel="RBF", jitter=1e-6, route="inv")
        np.testing.assert_allclose(mean[t], m_ref, rtol=1e-8)
        np.testing.assert_allclose(cov[t], c_ref, rtol=1e-6, atol=1e-9)
    mean1_, cov1_ = m.get_mvn_posterior(X_test, params, noiseless=False)
    mean2, cov2 = m.get_mvn_posterior(X_test, params, noiseless=True)
    np.testing.assert_array_equal(mean, mean1_)
    np.testing.assert_array_equal(cov, cov1_)
    np.testing.assert_array_equal(mean, mean2)
    assert np.count_nonzero(cov - cov2) > 0


@pytest.mark.parametrize("n", [1, 10])
def test_prediction(n):  # test_vgp.py:139-154
    X, y = get_dummy_data(unsqueeze=True)
    X_test, _ = get_dummy_data(unsqueeze=True, seed=1)
    rng = np.random.default_rng(0)
    samples = {"k_length": np.exp(0.3 * rng.standard_normal((20, 3, 1))), "k_scale": np.exp(0.3 * rng.standard_normal((20, 3))),
               "noise": 0.1 * np.exp(0.3 * rng.standard_normal((20, 3)))}
    m = vExactGP(1, "RBF")
    m.X_train, m.y_train = X, y
    y_mean, y_sampled = m.predict(get_keys()[1], X_test, samples, n=n)
    assert y_mean.shape == X_test.shape[:-1]
    assert y_sampled.shape == (20, n, *X_test.shape[:-1])
    # entry [s][t] is the posterior of task t at sample s
    s, t = 7, 2
    p = {"k_length": samples["k_length"][s, t], "k_scale": samples["k_scale"][s, t], "noise": samples["noise"][s, t]}
    m_ref, c_ref = ref.get_mvn_posterior(X[t], y[t], X_test[t], p, False, kernel="RBF", jitter=1e-6, route="inv")
    assert np.all(np.isfinite(y_sampled))
    # the mean over the n draws of one (s, t) fluctuates around the posterior mean with the posterior std
    z = (y_sampled[s, :, t, :].mean(0) - m_ref) / np.sqrt(np.diag(c_ref) / n)
    assert np.abs(z).max() < 6.0


@pytest.mark.parametrize("n", [1, 10])
def test_fit_predict_in_batches(n):  # test_vgp.py:157-184
    X, y = get_dummy_data()
    X_test, _ = get_dummy_data(seed=1)
    m = vExactGP(1, "RBF")
    m.fit(get_keys()[0], X, y, num_warmup=10, num_samples=10, progress_bar=False, print_summary=False)
    y_pred, y_sampled = m.predict_in_batches(get_keys()[1], X_test, batch_size=4, n=n)
    assert y_pred.shape == X_test.shape
    assert y_sampled.shape == (10, n, *X_test.shape)
    y_mean1, y_sampled1 = m.predict_in_batches(get_keys()[1], X_test, batch_size=4, n=n, noiseless=True)
    np.testing.assert_array_equal(y_mean1, y_pred)
    assert np.count_nonzero(y_sampled1 - y_sampled) > 0


def test_jitter_predict():  # test_vgp.py:187-200
    X, y = get_dummy_data(unsqueeze=True)
    X_test, _ = get_dummy_data(unsqueeze=True, seed=1)
    rng = np.random.default_rng(0)
    samples = {"k_length": np.exp(0.3 * rng.standard_normal((10, 3, 1))), "k_scale": np.exp(0.3 * rng.standard_normal((10, 3))),
               "noise": 0.1 * np.exp(0.3 * rng.standard_normal((10, 3)))}
    m = vExactGP(1, "RBF")
    m.X_train, m.y_train = X, y
    y_mean1, y_sampled1 = m.predict(get_keys()[1], X_test, samples, n=1, jitter=1e-6)
    y_mean2, y_sampled2 = m.predict(get_keys()[1], X_test, samples, n=1, jitter=1e-5)
    assert np.count_nonzero(y_sampled1 - y_sampled2) > 0
    assert np.count_nonzero(y_mean1 - y_mean2) > 0


def test_mean_function_with_prior():
    X, y = get_dummy_data()
    mean_fn = lambda x, p: p["a"] * x[..., 0] ** 2
    m = vExactGP(1, "RBF", mean_fn=mean_fn, mean_fn_prior={"a": dist.Normal(10.0, 1.0)})
    m.fit(get_keys()[0], X, y, num_warmup=15, num_samples=15, progress_bar=False, print_summary=False)
    s = m.get_samples()
    assert s["a"].shape == (15,)
    y_mean, y_sampled = m.predict(get_keys()[1], X, n=2)
    assert y_mean.shape == (3, 8) and y_sampled.shape == (15, 2, 3, 8)
    assert np.sqrt(np.mean((y_mean - y) ** 2)) < 2.0


def test_custom_kernel_and_noise_priors_are_used_not_ignored():
    """vExactGP(kernel_prior=..., noise_prior=...) must sample under THOSE priors (vgp.py:69-75), per task."""
    import gpax_amd
    from gpax_amd import dist, plate, sample
    from gpax_amd.models.vgp import vExactGP

    T, d = 3, 2

    def kprior():
        with plate("tasks", T, dim=-2):
            with plate("ard", d, dim=-1):
                length = sample("k_length", dist.Uniform(0.5, 2.0))
        with plate("tasks2", T):
            scale = sample("k_scale", dist.HalfNormal(3.0))
        return {"k_length": length, "k_scale": scale}

    def nprior():
        with plate("noise_plate", T):
            return sample("noise", dist.HalfNormal(0.1))

    m = vExactGP(d, "RBF", kernel_prior=kprior, noise_prior=nprior)
    m.X_train = np.zeros((T, 5, d))
    sites = {s.name: s for s in m._sites()}
    assert tuple(sites["k_length"].shape) == (T, d) and isinstance(sites["k_length"].dist, dist.Uniform)
    assert tuple(sites["k_scale"].shape) == (T,) and isinstance(sites["k_scale"].dist, dist.HalfNormal)
    assert tuple(sites["noise"].shape) == (T,) and isinstance(sites["noise"].dist, dist.HalfNormal)

    def scalar_prior():  # no task axis: the vmapped kernel could not consume it
        return {"k_length": sample("k_length", dist.LogNormal(0, 1)), "k_scale": sample("k_scale", dist.LogNormal(0, 1))}

    bad = vExactGP(d, "RBF", kernel_prior=scalar_prior)
    bad.X_train = np.zeros((T, 5, d))
    with pytest.raises(ValueError, match="tasks"):
        bad._sites()


@pytest.mark.parametrize("shape", ["T", "T1"])
def test_per_task_shared_lengthscale_gradient_matches_finite_differences(shape):
    """a `kernel_prior` whose k_length has shape (T,) or (T, 1) — one lengthscale per task, shared by the d
    input dimensions — must get, per task, the sum over ITS OWN d device-gradient entries (not the grand total broadcast
    to every task)."""
    from gpax_amd import dist, plate, sample
    from gpax_amd.models.vgp import vExactGP

    T, d, N = 3, 2, 12
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 3, (T, N, d))
    y = np.sin(X[..., 0]) * np.cos(X[..., 1]) + 0.1 * rng.standard_normal((T, N))

    def kprior():
        if shape == "T":
            with plate("tasks", T):
                length = sample("k_length", dist.LogNormal(0.0, 1.0))
        else:
            with plate("tasks", T, dim=-2):
                with plate("one", 1, dim=-1):
                    length = sample("k_length", dist.LogNormal(0.0, 1.0))
        with plate("tasks2", T):
            scale = sample("k_scale", dist.LogNormal(0.0, 1.0))
        return {"k_length": length, "k_scale": scale}

    m = vExactGP(d, "RBF", kernel_prior=kprior)
    m.X_train, m.y_train = m._set_data(X, y)
    sites = m._sites()
    u = 0.3 * rng.standard_normal(sum(s.size for s in sites))
    val, grad = m._log_joint(sites, u, 1e-6, jacobian=True)
    fd = np.zeros_like(u)
    h = 1e-6
    for i in range(u.size):
        up, um = u.copy(), u.copy()
        up[i] += h
        um[i] -= h
        fd[i] = (m._log_joint(sites, up, 1e-6, True)[0] - m._log_joint(sites, um, 1e-6, True)[0]) / (2 * h)
    np.testing.assert_allclose(grad, fd, rtol=2e-5, atol=2e-6)


def test_model_returns_the_sum_of_the_task_log_joints():
    """vExactGP.model(X, y) (vgp.py:62-96): per-task priors + the sum over tasks of the exact-GP log-likelihoods; y = None:
    the priors.  The inherited ExactGP.model knew nothing of the task axis."""
    X, y = get_dummy_data()
    m = vExactGP(1, "Matern")
    T = 3
    params = {"k_length": np.array([[0.7], [0.9], [1.1]]), "k_scale": np.array([1.2, 1.0, 0.8]), "noise": np.array([0.1, 0.2, 0.3])}
    ln = dist.LogNormal(0, 1)
    lp = ln.log_prob(params["k_length"].reshape(-1)).sum() + ln.log_prob(params["k_scale"]).sum() + ln.log_prob(params["noise"]).sum()
    assert abs(m.model(X, None, params=params) - lp) < 1e-12
    Xs = m._set_data(X)
    expect = sum(ref.exactgp_log_likelihood(Xs[t], y[t], {"k_length": params["k_length"][t], "k_scale": params["k_scale"][t],
                                                          "noise": params["noise"][t]}, kernel="Matern", jitter=1e-6)
                 for t in range(T))
    full = m.model(X, y, params=params)
    assert abs(full - (lp + expect)) < 1e-9 * abs(lp + expect)
    assert m.X_train is None and np.isfinite(m.model(X, y))
