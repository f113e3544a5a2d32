"""GPU: the Python model surface end to end on the real engine (plumbing config C1 of BASELINE.json:
ExactGP(1, 'RBF') on N=512 synthetic 1-D), parity with the oracle on deterministic quantities."""
import numpy as np
import pytest

from gpax_amd import _lib, dist
from gpax_amd.kernels import MaternKernel, RBFKernel
from gpax_amd.models import ExactGP, viGP
from gpax_amd.utils import get_keys
from oracle import cpu_ref as ref
import bench_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def real_engine(engine):
    _lib.set_engine(engine)
    yield
    _lib.set_engine(None)


def test_kernel_callables_on_gpu():
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 5, (40, 2))
    p = {"k_length": np.array([1.0, 2.0]), "k_scale": 1.5}
    np.testing.assert_allclose(RBFKernel(X, X, p, noise=0.1), ref.RBFKernel(X, X, p, noise=0.1), rtol=1e-10)
    np.testing.assert_allclose(MaternKernel(X, X[:7], p), ref.MaternKernel(X, X[:7], p), rtol=1e-10, atol=1e-12)
    # square_scaled_distance (kernels.py:28-41) by itself: the r^2 of the Gram kernels (GPX_KERNEL_R2), direct form — against
    # the reference's clipped expansion, and exactly 0 (not -1e-16 clipped) on the diagonal of X against itself
    from gpax_amd.kernels import square_scaled_distance
    for ell in (0.7, np.array([1.0, 2.0])):
        r2 = square_scaled_distance(X, X[:7], ell)
        np.testing.assert_allclose(r2, ref.square_scaled_distance(X, X[:7], ell), rtol=1e-10, atol=1e-12)
    assert np.all(np.diag(square_scaled_distance(X, X, 0.7)) == 0.0)
    Z5 = rng.uniform(0, 5, (33, 5))  # generic d
    np.testing.assert_allclose(square_scaled_distance(Z5, Z5[:4], 1.3), ref.square_scaled_distance(Z5, Z5[:4], 1.3),
                               rtol=1e-10, atol=1e-12)


def test_c1_exactgp_n512_fit_predict():
    X, y, Xn, p = bench_inputs.synthetic_problem(512, 1, 100, seed=0)
    m = ExactGP(1, "RBF")
    m.fit(get_keys()[0], X, y, num_warmup=60, num_samples=60, progress_bar=False, print_summary=False)
    s = m.get_samples()
    assert s["k_length"].shape == (60, 1)
    # the posterior concentrates near the generating noise level (0.1) for N=512
    assert 0.05 < np.median(s["noise"]) < 0.2
    ym, ys = m.predict(get_keys()[1], Xn, n=1)
    assert ym.shape == (100,) and ys.shape == (60, 1, 100) and np.isfinite(ys).all()
    f_true = np.sin(Xn[:, 0])
    assert np.sqrt(np.mean((ym - f_true) ** 2)) < 0.15
    # parity of the sweep with the oracle at the sampled thetas (deterministic given theta, eps)
    sub = {k: v[:4] for k, v in s.items()}
    mean_ref = np.stack([ref.get_mvn_posterior(X, y, Xn, {k: v[i] for k, v in sub.items()}, route="chol")[0]
                         for i in range(4)])
    ym4, _ = m.predict(get_keys()[1], Xn, samples=sub, n=1)
    np.testing.assert_allclose(ym4, mean_ref.mean(0), rtol=1e-7, atol=1e-9)


def test_log_joint_gradient_matches_finite_differences_on_gpu():
    X, y, _, _ = bench_inputs.synthetic_problem(150, 2, 4, seed=2)
    m = ExactGP(2, "Matern", noise_prior_dist=dist.HalfNormal(0.5))
    m.X_train, m.y_train = m._set_data(X, y)
    sites = m._sites()
    u = np.array([0.1, -0.3, 0.2, -1.5])
    v, g = m._log_joint(sites, u, 1e-6, True)
    for i in range(4):
        h = 1e-5
        up, um = u.copy(), u.copy()
        up[i] += h
        um[i] -= h
        fd = (m._log_joint(sites, up, 1e-6, True, False)[0] - m._log_joint(sites, um, 1e-6, True, False)[0]) / (2 * h)
        assert abs(fd - g[i]) <= 1e-6 * max(1.0, abs(fd))


@pytest.mark.parametrize("guide", ["delta", "normal"])
def test_vigp_on_gpu(guide):
    X, y, Xn, _ = bench_inputs.synthetic_problem(300, 2, 50, seed=4)
    m = viGP(2, "Matern", guide=guide)
    m.fit(get_keys()[0], X, y, num_steps=100, step_size=0.05, progress_bar=False, print_summary=False)
    s = m.get_samples()
    mean, var = m.predict(get_keys()[1], Xn)
    m_ref, v_ref = ref.vigp_predict(X, y, Xn, s, kernel="Matern")
    np.testing.assert_allclose(mean, m_ref, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, v_ref, rtol=1e-7, atol=1e-9)
    mb, vb = m.predict_in_batches(get_keys()[1], Xn, batch_size=16)
    np.testing.assert_allclose(mb, mean, rtol=1e-12)
    np.testing.assert_allclose(vb, var, rtol=1e-12)
    if guide == "delta":
        assert m.loss[-1] < m.loss[0]


def test_get_mvn_posterior_on_gpu_matches_reference_inverse_route():
    X, y, Xn, p = bench_inputs.synthetic_problem(200, 2, 30, seed=6)
    m = ExactGP(2, "RBF")
    m.X_train, m.y_train = m._set_data(X, y)
    params = {"k_length": p["k_length"], "k_scale": np.array([p["k_scale"]]), "noise": np.array([p["noise"]])}
    mean, cov = m.get_mvn_posterior(Xn, params, noiseless=True, jitter=1e-5)
    m_ref, c_ref = ref.get_mvn_posterior(X, y, Xn, p, True, kernel="RBF", jitter=1e-5, route="inv")
    np.testing.assert_allclose(mean, m_ref, rtol=1e-8)
    np.testing.assert_allclose(cov, c_ref, rtol=1e-6, atol=1e-9)
    ymean, ydraw = m._predict(get_keys()[1], Xn, params, 3)
    assert ydraw.shape == (3, 30)


@pytest.mark.parametrize("guide", ["delta", "normal"])
def test_visparsegp_on_gpu(guide):
    from gpax_amd.models import viSparseGP
    from gpax_amd.utils import preprocess_sparse_image

    rng = np.random.default_rng(3)
    img = np.fromfunction(lambda i, j: np.sin(i / 6.0) * np.cos(j / 5.0) + 1.5, (48, 48))
    sparse = img * (rng.uniform(size=img.shape) < 0.25)
    X, y, X_full = preprocess_sparse_image(sparse)
    m = viSparseGP(2, "Matern", guide=guide)
    m.fit(get_keys()[0], X, y, inducing_points_ratio=0.2, num_steps=60, step_size=0.05, progress_bar=False,
          print_summary=False)
    assert m.Xu.shape == (int(len(X) * 0.2), 2)
    mean, var = m.predict_in_batches(get_keys()[1], X_full, batch_size=1000)
    assert mean.shape == (48 * 48,) and var.shape == (48 * 48,) and np.all(var > 0)
    s = m.get_samples()
    m_ref, c_ref = ref.sparse_posterior(X, y, m.Xu, X_full[:200], {k: np.asarray(v) for k, v in s.items()},
                                        kernel="Matern")
    np.testing.assert_allclose(mean[:200], m_ref, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(var[:200], np.diag(c_ref), rtol=1e-6, atol=1e-7)
    if guide == "delta":
        assert m.loss[-1] < m.loss[0]
        rmse = np.sqrt(np.mean((mean - img.reshape(-1)) ** 2))
        assert rmse < 0.25


def test_parallel_chains_on_gpu_match_sequential():
    import time
    X, y, _, _ = bench_inputs.synthetic_problem(256, 1, 4, seed=1)
    out, dt = [], []
    for method in ["sequential", "parallel"]:
        m = ExactGP(1, "RBF")
        t0 = time.perf_counter()
        m.fit(get_keys()[0], X, y, num_warmup=30, num_samples=30, num_chains=3, chain_method=method,
              progress_bar=False, print_summary=False)
        dt.append(time.perf_counter() - t0)
        out.append(m.get_samples(chain_dim=True))
    for k in out[0]:
        np.testing.assert_array_equal(out[0][k], out[1][k])  # own generator per chain; lockstep-batched fit steps are bit-identical to single ones
    print(f"3 chains N=256: sequential {dt[0]:.2f} s, parallel {dt[1]:.2f} s")


def test_predict_with_threefry_keys_on_gpu():
    # utils.threefry keys: predict's eps are split(key, S) -> normal(key_s, (n, M)), as the reference draws them
    from gpax_amd.utils import threefry as tf
    X, y, Xn, _ = bench_inputs.synthetic_problem(120, 1, 9, seed=2)
    k1, k2 = tf.get_keys(0)
    m = ExactGP(1, "Matern")
    m.fit(k1, X, y, num_warmup=20, num_samples=12, progress_bar=False, print_summary=False)
    ym, ys = m.predict(k2, Xn, n=2)
    s = m.get_samples()
    eps = tf.predict_normals(k2, 12, 2, 9)
    for i in (0, 5, 11):
        p = {k: v[i] for k, v in s.items()}
        m_ref, c_ref = ref.get_mvn_posterior(X, y, Xn, {"k_length": p["k_length"], "k_scale": float(p["k_scale"]),
                                                        "noise": float(p["noise"])}, False, kernel="Matern", route="inv")
        np.testing.assert_allclose(ys[i], ref.mvn_sample(m_ref, c_ref, eps[i]), rtol=1e-6, atol=1e-8)


def test_sample_from_prior_on_gpu_matches_the_oracle_mvn_sample():
    """A15 (gp.py:401-408): theta ~ priors, y ~ MVN(mean, K(theta)); device Gram + device Cholesky per draw against
    ref.mvn_sample given the same generator (sites in order, then N standard normals per draw)."""
    from gpax_amd.utils.utils import rng_from_key
    for kernel, kfn in [("RBF", ref.RBFKernel), ("Matern", ref.MaternKernel)]:
        X = np.random.default_rng(1).uniform(0, 4, (150, 2))
        m = ExactGP(2, kernel, noise_prior_dist=dist.HalfNormal(0.3))
        out = m.sample_from_prior(11, X, num_samples=4)
        assert out.shape == (4, 150) and np.isfinite(out).all()
        rng = rng_from_key(11)
        for i in range(4):
            theta = {s.name: (s.dist.sample(rng, s.shape) if s.shape else float(s.dist.sample(rng))) for s in m._sites()}
            eps = rng.standard_normal(150)
            K = kfn(X, X, theta, theta["noise"], jitter=1e-6)
            expect = ref.mvn_sample(np.zeros(150), K, eps[None])[0]
            assert np.linalg.norm(out[i] - expect) <= 1e-8 * np.linalg.norm(expect)


def test_model_log_joint_on_gpu():
    X, y, _, p = bench_inputs.synthetic_problem(700, 2, 4, seed=4)
    m = ExactGP(2, "Matern")
    params = {"k_length": p["k_length"], "k_scale": p["k_scale"], "noise": p["noise"]}
    prior = m.model(X, None, params=params)
    full = m.model(X, y, params=params)
    expect = ref.exactgp_log_likelihood(X, y, p, kernel="Matern")
    assert abs((full - prior) - expect) <= 1e-10 * abs(expect)


@pytest.mark.parametrize("kernel", ["RBF", "Matern"])
def test_sparse_model_log_joint_on_gpu(kernel):
    """viSparseGP.model(X, y, Xu) (sparse_gp.py:62-114): priors + LowRankMVN likelihood - trace_term / 2 = the VFE bound
    (gpx_sgp_bound) + the site log-densities; a lower bound of the exact log joint at the same theta; y = None: priors."""
    from gpax_amd.models import viSparseGP
    X, y, _, p = bench_inputs.synthetic_problem(700, 2, 4, seed=4)
    Xu = X[np.random.default_rng(5).choice(700, 90, replace=False)] + 0.01
    params = {"k_length": p["k_length"], "k_scale": p["k_scale"], "noise": p["noise"]}
    m = viSparseGP(2, kernel, noise_prior_dist=dist.HalfNormal(0.5))
    lp = dist.LogNormal(0, 1).log_prob(np.asarray(p["k_length"], dtype=float)).sum() \
        + dist.LogNormal(0, 1).log_prob(np.array([p["k_scale"]]))[0] + dist.HalfNormal(0.5).log_prob(np.array([p["noise"]]))[0]
    assert abs(m.model(X, None, params=params) - lp) < 1e-12
    full = m.model(X, y, Xu, params=params)
    expect = ref.sparse_bound(X, y, Xu, p, kernel=kernel, jitter=1e-6)
    assert abs((full - lp) - expect) <= 1e-9 * abs(expect)
    exact = ExactGP(2, kernel, noise_prior_dist=dist.HalfNormal(0.5)).model(X, y, params=params)
    assert full <= exact
    m.Xu = Xu  # what fit() stores: the default of the Xu argument
    assert m.model(X, y, params=params) == full
    assert np.isfinite(m.model(X, y, Xu))  # default params: the prior medians
    assert np.isnan(m.model(X, y, Xu, params={**params, "k_scale": -1.0}))
    with pytest.raises(ValueError):
        viSparseGP(2, kernel).model(X, y)


def test_custom_kernel_prior_fit_on_gpu():  # gpax/tests/test_gp.py:129-134
    import gpax_amd

    def prior():
        length = gpax_amd.sample("k_length", dist.Uniform(0, 1))
        scale = gpax_amd.sample("k_scale", dist.LogNormal(0, 1))
        return {"k_length": length, "k_scale": scale}

    rng = np.random.default_rng(0)
    X = np.linspace(1, 2, 8) + 0.01 * rng.standard_normal(8)
    y = 10 * X ** 2
    with pytest.warns(UserWarning):
        m = ExactGP(1, "RBF", kernel_prior=prior)
    m.fit(get_keys()[0], X, y, num_warmup=30, num_samples=30, progress_bar=False, print_summary=False)
    s = m.get_samples()
    assert s["k_length"].shape == (30,) and np.all((s["k_length"] > 0) & (s["k_length"] < 1))
