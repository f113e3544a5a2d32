"""CPU: the host-side model classes (ExactGP / viGP surface, NUTS / SVI drivers, plumbing) run
against a checker-backed engine injected from tests/oracle_engine.py.  Mirrors the reference's
own assertions (gpax/tests/test_gp.py, test_vigp.py, test_utils.py, test_kernels.py): shapes,
types, determinism, sensitivity — plus parity of deterministic quantities with the oracle."""
import numpy as np
import pytest

import gpax_amd
from gpax_amd import _lib, dist
from gpax_amd.models import ExactGP, viGP, viSparseGP
from gpax_amd.utils import (get_keys, initialize_inducing_points, preprocess_sparse_image, random_sample_dict,
                            split_dict, split_in_batches)
from oracle import cpu_ref as ref
import bench_inputs
from tests.oracle_engine import OracleEngine


@pytest.fixture(autouse=True)
def oracle_engine():
    eng = OracleEngine()
    _lib.set_engine(eng)
    yield eng
    _lib.set_engine(None)


def get_dummy_data(n=8, seed=0):
    rng = np.random.default_rng(seed)
    X = np.linspace(1, 2, n) + 0.1 * rng.standard_normal(n)
    y = 10 * X ** 2 * 0.1
    return X, y


@pytest.mark.parametrize("kernel", ["RBF", "Matern"])
@pytest.mark.parametrize("dim", [1, 2])
def test_fit_get_samples_shapes(kernel, dim):
    rng_key = get_keys()[0]
    X, y = get_dummy_data()
    if dim == 2:
        X = np.stack([X, X[::-1]], 1)
    m = ExactGP(dim, kernel)
    m.fit(rng_key, X, y, num_warmup=30, num_samples=40, progress_bar=False, print_summary=False)
    assert m.mcmc is not None
    s = m.get_samples()
    assert set(s) == {"k_length", "k_scale", "noise"}
    assert s["k_length"].shape == (40, dim) and s["k_scale"].shape == (40,) and s["noise"].shape == (40,)
    assert all(np.all(v > 0) for v in s.values())
    s2 = m.get_samples(chain_dim=True)
    assert s2["k_length"].shape == (1, 40, dim)


def test_fit_same_key_same_samples_and_jitter_sensitivity():
    X, y = get_dummy_data()
    outs = []
    for jit in [1e-6, 1e-6, 1e-5]:
        m = ExactGP(1, "RBF")
        m.fit(get_keys()[0], X, y, num_warmup=20, num_samples=20, progress_bar=False, print_summary=False, jitter=jit)
        outs.append(m.get_samples()["k_length"])
    np.testing.assert_array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])


def test_two_chains_and_custom_priors():
    X, y = get_dummy_data()
    m = ExactGP(1, "Matern", noise_prior_dist=dist.HalfNormal(0.1), lengthscale_prior_dist=dist.Gamma(2, 5))
    m.fit(get_keys()[0], X, y, num_warmup=20, num_samples=25, num_chains=2, progress_bar=False, print_summary=True)
    assert m.get_samples(chain_dim=True)["k_length"].shape == (2, 25, 1)
    assert m.get_samples()["noise"].shape == (50,)
    assert np.median(m.get_samples()["noise"]) < 1.0


def test_unsupported_inputs_raise_clearly():
    with pytest.raises(NotImplementedError):
        ExactGP(1, "NNGP")
    with pytest.raises(NotImplementedError):
        ExactGP(1, lambda a, b, c: None)
    with pytest.warns(UserWarning):
        m = ExactGP(1, "RBF", kernel_prior=lambda: {})
    with pytest.raises(NotImplementedError):  # a prior callable that registers no gpax_amd.sample site
        m._sites()


@pytest.mark.parametrize("noiseless", [False, True])
def test_get_mvn_posterior_matches_oracle_and_invariants(noiseless):
    X, y, Xn, p = bench_inputs.synthetic_problem(40, 2, 12, seed=3)
    m = ExactGP(2, "RBF")
    m.X_train, m.y_train = m._set_data(X, y)
    params = {"k_length": np.array([[1.0, 1.25]]), "k_scale": np.array([1.3]), "noise": np.array([0.1])}
    mean, cov = m.get_mvn_posterior(Xn, params, noiseless)
    assert mean.shape == (12,) and cov.shape == (12, 12)
    m_ref, c_ref = ref.get_mvn_posterior(X, y, Xn, p, noiseless, kernel="RBF", route="inv")
    np.testing.assert_allclose(mean, m_ref, rtol=1e-9)
    np.testing.assert_allclose(cov, c_ref, rtol=1e-8, atol=1e-10)
    mean2, cov2 = m.get_mvn_posterior(Xn, params, noiseless)
    np.testing.assert_array_equal(mean, mean2)
    np.testing.assert_array_equal(cov, cov2)


def test_predict_shapes_negative_variances_do_not_crash():
    # gpax/tests/test_gp.py:173-206: "samples" drawn from N(0,1)
    X, y = get_dummy_data()
    rng = np.random.default_rng(1)
    samples = {"k_length": rng.standard_normal((100, 1)), "k_scale": rng.standard_normal(100),
               "noise": rng.standard_normal(100)}
    samples["k_length"] = np.abs(samples["k_length"]) + 0.1
    m = ExactGP(1, "RBF")
    m.X_train, m.y_train = m._set_data(X, y)
    Xn = np.linspace(1, 2, 5)
    for n in [1, 10]:
        y_mean, y_sampled = m.predict(get_keys()[1], Xn, samples, n=n)
        assert y_mean.shape == (5,) and y_sampled.shape == (100, n, 5)
    _, ys = m.predict(get_keys()[1], Xn, samples, n=1, filter_nans=True)
    assert ys.shape[0] < 100 and not np.isnan(ys).any()


def test_predict_in_batches_equals_predict_on_means():
    X, y, Xn, _ = bench_inputs.synthetic_problem(30, 1, 8, seed=5)
    samples = bench_inputs.synthetic_theta_samples(6, 1)
    m = ExactGP(1, "Matern")
    m.X_train, m.y_train = m._set_data(X, y)
    key = get_keys()[1]
    ym, ys = m.predict(key, Xn, samples, n=2)
    for bs in [2, 3, 8]:
        ymb, ysb = m.predict_in_batches(key, Xn, bs, samples, n=2)
        assert ymb.shape == (8,) and ysb.shape == (6, 2, 8)
        np.testing.assert_allclose(ymb, ym, rtol=1e-9)


def test_predict_with_mean_function_and_prior():
    X, y, Xn, _ = bench_inputs.synthetic_problem(30, 1, 8, seed=6)
    y = y + 2.0 * X[:, 0]
    mean_fn = lambda x, p: p["a"] * x[:, 0]
    m = ExactGP(1, "RBF", mean_fn=mean_fn, mean_fn_prior={"a": dist.Normal(2.0, 0.5)})
    m.fit(get_keys()[0], X, y, num_warmup=30, num_samples=30, progress_bar=False, print_summary=False)
    s = m.get_samples()
    assert "a" in s and s["a"].shape == (30,) and abs(np.median(s["a"]) - 2.0) < 1.0
    ym, ys = m.predict(get_keys()[1], Xn, n=1)
    assert ym.shape == (8,) and ys.shape == (30, 1, 8)
    # parity of the mean-function chain rule with finite differences of the log joint
    sites = m._sites()
    u = np.array([0.1, -0.2, -1.0, 1.9])
    v, g = m._log_joint(sites, u, 1e-6, True)
    for i in range(4):
        h = 1e-6
        up, um = u.copy(), u.copy()
        up[i] += h
        um[i] -= h
        fd = (m._log_joint(sites, up, 1e-6, True, False)[0] - m._log_joint(sites, um, 1e-6, True, False)[0]) / (2 * h)
        assert abs(fd - g[i]) <= 1e-5 * max(1.0, abs(fd))


def test_sample_from_prior_shape():
    m = ExactGP(1, "RBF")
    out = m.sample_from_prior(get_keys()[0], np.linspace(0, 1, 9), num_samples=4)
    assert out.shape == (4, 9)


@pytest.mark.parametrize("guide", ["delta", "normal"])
def test_vigp_fit_predict(guide):
    X, y, Xn, p = bench_inputs.synthetic_problem(40, 1, 10, seed=8)
    m = viGP(1, "Matern", guide=guide)
    m.fit(get_keys()[0], X, y, num_steps=60, step_size=0.05, progress_bar=False, print_summary=(guide == "delta"))
    assert m.svi is not None
    s = m.get_samples()
    assert set(s) == {"k_length", "k_scale", "noise"} and s["k_length"].shape == (1,)
    mean, var = m.predict(get_keys()[1], Xn)
    assert mean.shape == (10,) and var.shape == (10,) and np.all(var > 0)
    m_ref, v_ref = ref.vigp_predict(X, y, Xn, {k: np.asarray(v) for k, v in s.items()}, kernel="Matern")
    np.testing.assert_allclose(mean, m_ref, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(var, v_ref, rtol=1e-7, atol=1e-10)
    mb, vb = m.predict_in_batches(get_keys()[1], Xn, batch_size=3)
    np.testing.assert_allclose(mb, mean, rtol=1e-10)
    np.testing.assert_allclose(vb, var, rtol=1e-10)
    if guide == "delta":  # MAP improves the objective
        assert np.nanmin(m.loss[-10:]) < m.loss[0]
    # broadcastable parameter shapes (gpax/tests/test_vigp.py:68-99)
    params = {"k_length": np.array([[1.0]]), "k_scale": np.array([1.0]), "noise": np.array([0.1])}
    mean2, var2 = m.predict(get_keys()[1], Xn, samples=params)
    assert mean2.shape == (10,) and var2.shape == (10,)


def test_vigp_same_key_identical_and_guides_differ():
    X, y, Xn, _ = bench_inputs.synthetic_problem(25, 1, 5, seed=9)
    res = []
    for guide in ["delta", "delta", "normal"]:
        m = viGP(1, "RBF", guide=guide)
        m.fit(get_keys()[0], X, y, num_steps=20, progress_bar=False, print_summary=False)
        res.append(m.get_samples()["k_length"])
    np.testing.assert_array_equal(res[0], res[1])
    assert not np.array_equal(res[0], res[2])


def test_kernel_callables_follow_the_protocol():
    from gpax_amd.kernels import MaternKernel, RBFKernel, get_kernel
    rng = np.random.default_rng(0)
    x1, x2 = rng.standard_normal((5, 2)), rng.standard_normal((5, 2))
    for k in [RBFKernel, MaternKernel]:
        for ell in [1.0, np.array([1.0, 2.0]), np.array([1.0]), np.array([[1.0]])]:
            K = k(x1, x2, {"k_length": ell, "k_scale": 1.0})
            assert isinstance(K, np.ndarray) and K.shape == (5, 5)
    assert get_kernel("RBF") is RBFKernel and get_kernel(MaternKernel) is MaternKernel
    with pytest.raises(KeyError):
        get_kernel("Nope")
    K = RBFKernel(x1, x1, {"k_length": 1.0, "k_scale": 1.0}, noise=0.3, jitter=1e-6)
    Kr = ref.RBFKernel(x1, x1, {"k_length": 1.0, "k_scale": 1.0}, noise=0.3, jitter=1e-6)
    np.testing.assert_allclose(K, Kr, rtol=1e-12)
    # square_scaled_distance (kernels.py:28-41): (n, m), scalar / ARD lengthscale, 1-D inputs, never negative
    from gpax_amd.kernels import square_scaled_distance
    r2 = square_scaled_distance(x1, x2[:3], np.array([1.0, 2.0]))
    assert r2.shape == (5, 3) and np.all(r2 >= 0)
    np.testing.assert_allclose(r2, ref.square_scaled_distance(x1, x2[:3], np.array([1.0, 2.0])), rtol=1e-12, atol=1e-14)
    assert square_scaled_distance(x1[:, 0], x1[:, 0], 0.5).shape == (5, 5)


def test_utils_match_reference_behaviour():
    g = np.load("tests/golden/utils.npz")
    for tag in ["img34", "img16"]:
        a, b, c = preprocess_sparse_image(g[tag])
        np.testing.assert_array_equal(a, g[tag + "_X"])
        np.testing.assert_array_equal(b, g[tag + "_y"])
        np.testing.assert_array_equal(c, g[tag + "_full"])
    assert [len(q) for q in split_in_batches(np.arange(10), 4)] == [4, 4, 2]
    assert [q.shape for q in split_in_batches(np.zeros((2, 7)), 3, dim=1)] == [(2, 3), (2, 3), (2, 1)]
    assert len(split_in_batches(np.arange(3), 4)) == 1
    with pytest.raises(NotImplementedError):
        split_in_batches(np.arange(3), 2, dim=2)
    d = {"a": np.arange(10), "b": np.arange(20).reshape(10, 2)}
    assert [len(q["a"]) for q in split_dict(d, 4)] == [4, 4, 2]
    sub = random_sample_dict(d, 3, get_keys()[0])
    assert sub["a"].shape == (3,) and np.array_equal(sub["b"][:, 0] // 2, sub["a"])
    X = np.arange(40.0).reshape(20, 2)
    assert initialize_inducing_points(X, 0.25, "uniform").shape == (5, 2)
    assert initialize_inducing_points(X, 0.25, "random", key=get_keys()[0]).shape == (5, 2)
    with pytest.raises(ValueError):
        initialize_inducing_points(X, 1.5)
    with pytest.raises(ValueError):
        initialize_inducing_points(X, 0.5, "random")
    with pytest.raises(ValueError):
        initialize_inducing_points(X, 0.5, "bogus")
    k1, k2 = get_keys(1)
    assert not np.array_equal(k1, k2) and np.array_equal(get_keys(1)[0], k1)
    assert gpax_amd.utils.enable_x64() is None


@pytest.mark.parametrize("guide", ["delta", "normal"])
def test_visparsegp_fit_predict(guide):
    # gpax/tests/test_sparsegp.py:26-64: Xu is an array, is optimised, posterior shapes
    X, y, Xn, _ = bench_inputs.synthetic_problem(24, 1, 7, seed=4)
    key = get_keys()[0]
    m1 = viSparseGP(1, "Matern", guide=guide)
    m1.fit(key, X, y, inducing_points_ratio=0.2, num_steps=1, progress_bar=False, print_summary=False)
    m2 = viSparseGP(1, "Matern", guide=guide)
    m2.fit(key, X, y, inducing_points_ratio=0.2, num_steps=25, step_size=0.05, progress_bar=False, print_summary=False)
    assert isinstance(m2.Xu, np.ndarray) and m2.Xu.shape == (4, 1)
    assert not np.allclose(m1.Xu, m2.Xu)
    mean, var = m2.predict(get_keys()[1], Xn)
    assert mean.shape == (7,) and var.shape == (7,)
    mean2, cov = m2.get_mvn_posterior(Xn, m2.get_samples())
    assert cov.shape == (7, 7)
    np.testing.assert_allclose(mean, mean2, rtol=1e-10)
    np.testing.assert_allclose(var, np.diag(cov), rtol=1e-8)
    mb, vb = m2.predict_in_batches(get_keys()[1], Xn, batch_size=3)
    np.testing.assert_allclose(mb, mean, rtol=1e-10)
    if guide == "delta":
        assert np.nanmin(m2.loss[-5:]) < m2.loss[0]


def test_visparsegp_vector_mean_site_gradient_and_posterior_mean():
    """VERDICT r3 item 7 / sparse_gp.py:85-89,189-192,219-221: a mean function whose prior holds a 2-element site.  The
    objective's gradient must carry one entry per ELEMENT (central differences of the objective itself), and the
    posterior is the oracle's Woodbury posterior of the residual with the mean added back at X_new."""
    from gpax_amd import dist, plate, sample

    X, y, Xn, _ = bench_inputs.synthetic_problem(30, 1, 6, seed=8)
    y = y + 0.7 * X[:, 0] - 0.2

    def mean_fn(x, p):
        return p["w"][0] * x[:, 0] + p["w"][1]

    def mean_prior():
        with plate("w_plate", 2):
            w = sample("w", dist.Normal(0.0, 2.0))
        return {"w": w}

    m = viSparseGP(1, "Matern", mean_fn=mean_fn, mean_fn_prior=mean_prior)
    m.X_train, m.y_train = m._set_data(X, y)
    sites = m._sites()
    assert {s.name: tuple(s.shape) for s in sites}["w"] == (2,)
    rng = np.random.default_rng(0)
    nu = sum(s.size for s in sites)
    Xu = X[rng.choice(30, 5, replace=False)].copy()
    x0 = np.concatenate([0.2 * rng.standard_normal(nu), Xu.reshape(-1)])
    val, grad = m._sparse_log_joint(sites, x0, 5, 1e-5, jacobian=False)
    off = 0
    for s_ in sites:
        if s_.name == "w":
            break
        off += s_.size
    for i in (off, off + 1):
        h = 1e-5
        xp, xm = x0.copy(), x0.copy()
        xp[i] += h
        xm[i] -= h
        fd = (m._sparse_log_joint(sites, xp, 5, 1e-5, False)[0] - m._sparse_log_joint(sites, xm, 5, 1e-5, False)[0]) / (2 * h)
        assert abs(grad[i] - fd) <= 1e-5 * max(1.0, abs(fd)), (i, grad[i], fd)
    assert abs(grad[off] - grad[off + 1]) > 1e-6  # two different derivatives, not one value broadcast
    m.Xu = Xu
    params = {"k_length": np.array([1.3]), "k_scale": 1.1, "noise": 0.05, "w": np.array([0.6, -0.1])}
    mean, cov = m.get_mvn_posterior(Xn, params, jitter=1e-5)
    e_mean, e_cov = ref.sparse_posterior(X, y, Xu, Xn, params, False, kernel="Matern", jitter=1e-5, mean_fn=mean_fn,
                                         mean_fn_has_params=True)
    np.testing.assert_allclose(mean, e_mean, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(cov, e_cov, rtol=1e-8, atol=1e-11)


def test_parallel_chains_equal_sequential_chains():
    # chains own independent generators, so the schedule does not change the draws
    X, y = get_dummy_data()
    outs = []
    for method in ["sequential", "parallel"]:
        m = ExactGP(1, "RBF")
        m.fit(get_keys()[0], X, y, num_warmup=15, num_samples=15, num_chains=2, chain_method=method,
              progress_bar=False, print_summary=False)
        outs.append(m.get_samples(chain_dim=True)["k_length"])
    assert outs[0].shape == (2, 15, 1)
    np.testing.assert_array_equal(outs[0], outs[1])


def test_periodic_kernel_model_surface():
    # gpax/tests/test_gp.py:41-49 parametrises 'Periodic' as well: fit + predict run, 'period' is a sample site
    from gpax_amd.kernels import PeriodicKernel
    rng = np.random.default_rng(0)
    X = np.linspace(0, 6, 24)
    y = np.sin(2 * np.pi * X / 2.0) + 0.05 * rng.standard_normal(24)
    K = PeriodicKernel(X[:, None], X[:, None], {"k_length": 1.0, "k_scale": 1.0, "period": 2.0}, noise=0.1)
    Kr = ref.PeriodicKernel(X[:, None], X[:, None], {"k_length": 1.0, "k_scale": 1.0, "period": 2.0}, noise=0.1)
    np.testing.assert_allclose(K, Kr, rtol=1e-12)
    m = ExactGP(1, "Periodic")
    m.fit(get_keys()[0], X, y, num_warmup=20, num_samples=20, progress_bar=False, print_summary=False)
    s = m.get_samples()
    assert set(s) == {"k_length", "k_scale", "period", "noise"} and s["period"].shape == (20,)
    ym, ys = m.predict(get_keys()[1], np.linspace(0, 6, 7), n=1)
    assert ym.shape == (7,) and ys.shape == (20, 1, 7)
    v = viGP(1, "Periodic")
    v.fit(get_keys()[0], X, y, num_steps=15, progress_bar=False, print_summary=False)
    mean, var = v.predict(get_keys()[1], np.linspace(0, 6, 7))
    assert mean.shape == (7,) and np.all(var > 0)
    with pytest.raises(NotImplementedError):
        viSparseGP(1, "Periodic")


def test_predict_in_batches_equals_slice_by_slice_predict():
    # gp.py:325-349: predict per slice of X_new with the SAME key; the default path runs them as covariance blocks of
    # one sweep (one factorisation per sample) and must return exactly what the slice-by-slice loop returns
    X, y, Xn, p = bench_inputs.synthetic_problem(40, 2, 23, seed=3)
    rng = np.random.default_rng(0)
    samples = {"k_length": np.exp(0.2 * rng.standard_normal((6, 2))), "k_scale": np.exp(0.2 * rng.standard_normal(6)),
               "noise": 0.1 * np.exp(0.2 * rng.standard_normal(6))}
    m = ExactGP(2, "Matern")
    m.X_train, m.y_train = m._set_data(X, y)
    key = get_keys()[1]
    ym, ys = m.predict_in_batches(key, Xn, batch_size=10, samples=samples, n=3)
    assert ym.shape == (23,) and ys.shape == (6, 3, 23)
    parts = [m.predict(key, Xn[i:i + 10], samples, n=3) for i in range(0, 23, 10)]
    np.testing.assert_allclose(ym, np.concatenate([q[0] for q in parts]), rtol=1e-12)
    np.testing.assert_allclose(ys, np.concatenate([q[1] for q in parts], axis=-1), rtol=1e-12)
    # a user-supplied predict_fn still goes slice by slice
    calls = []
    fn = lambda xi: (calls.append(len(xi)) or m.predict(key, xi, samples, n=1))
    m.predict_in_batches(key, Xn, batch_size=10, samples=samples, predict_fn=fn)
    assert calls == [10, 10, 3]


def test_predict_in_batches_groups_of_slices_give_the_same_values(monkeypatch):
    X, y, Xn, p = bench_inputs.synthetic_problem(30, 1, 47, seed=5)
    rng = np.random.default_rng(1)
    samples = {"k_length": np.exp(0.2 * rng.standard_normal((4, 1))), "k_scale": np.exp(0.2 * rng.standard_normal(4)),
               "noise": 0.1 * np.exp(0.2 * rng.standard_normal(4))}
    m = ExactGP(1, "RBF")
    m.X_train, m.y_train = m._set_data(X, y)
    key = get_keys()[1]
    a = m.predict_in_batches(key, Xn, batch_size=10, samples=samples, n=2)
    monkeypatch.setattr(ExactGP, "_ride_along_bytes", 8.0 * (30 + 256) * 20)  # two slices per sweep
    b = m.predict_in_batches(key, Xn, batch_size=10, samples=samples, n=2)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_predict_of_one_model_does_not_leave_another_model_stale_training_inputs():
    """with a single context, A.predict() re-uploaded X while the engine still recorded B as owner;
    B's next posterior then ran on A's training inputs.  Engine.set_train now drops the owner itself."""
    from gpax_amd import ExactGP
    rng = np.random.default_rng(0)
    XA, XB = rng.uniform(0, 3, (12, 1)), rng.uniform(0, 3, (12, 1))
    yA, yB = np.sin(XA[:, 0]), np.cos(2 * XB[:, 0])
    Xn = np.linspace(0, 3, 7)[:, None]
    p = {"k_length": np.array([0.7]), "k_scale": 1.1, "noise": 0.05}
    A, B = ExactGP(1, "RBF"), ExactGP(1, "RBF")
    A.X_train, A.y_train = A._set_data(XA, yA)
    B.X_train, B.y_train = B._set_data(XB, yB)
    mB0, _ = B.get_mvn_posterior(Xn, p)
    samples = {k: np.asarray([v, v]) for k, v in p.items()}
    A.predict(0, Xn, samples=samples, n=1)
    mB1, _ = B.get_mvn_posterior(Xn, p)
    np.testing.assert_array_equal(mB0, mB1)


def test_refit_with_a_mutated_X_array_reuploads_the_training_inputs():
    from gpax_amd import viGP
    rng = np.random.default_rng(1)
    X = np.ascontiguousarray(rng.uniform(0, 3, (10, 1)))
    y = np.sin(X[:, 0])
    m = viGP(1, "RBF")
    m.fit(0, X, y, num_steps=2, progress_bar=False, print_summary=False)
    eng = _lib.get_engine()
    X[:] = X + 1.0  # same object, new contents
    m.fit(0, X, y, num_steps=2, progress_bar=False, print_summary=False)
    np.testing.assert_array_equal(eng.X, X)


# ---- prior callables: the reference's numpyro.sample programs with the import swapped (gp.py:96-135) ----------

def dummy_mean_fn(x, params):  # gpax/tests/test_gp.py:25-26
    return params["a"] * x[:, 0] ** params["b"]


def dummy_mean_fn_priors():  # gpax/tests/test_gp.py:29-32
    import gpax_amd
    a = gpax_amd.sample("a", dist.LogNormal(0, 1))
    b = gpax_amd.sample("b", dist.Normal(3, 1))
    return {"a": a, "b": b}


def gp_kernel_custom_prior():  # gpax/tests/test_gp.py:35-38
    import gpax_amd
    length = gpax_amd.sample("k_length", dist.Uniform(0, 1))
    scale = gpax_amd.sample("k_scale", dist.LogNormal(0, 1))
    return {"k_length": length, "k_scale": scale}


def _dummy(n=8, seed=0):
    rng = np.random.default_rng(seed)
    X = np.linspace(1, 2, n) + 0.01 * rng.standard_normal(n)
    return X, 10 * X ** 2


@pytest.mark.parametrize("kernel", ["RBF", "Matern"])
def test_fit_with_custom_kernel_priors(kernel):  # gpax/tests/test_gp.py:129-134
    X, y = _dummy()
    with pytest.warns(UserWarning):
        m = ExactGP(1, kernel, kernel_prior=gp_kernel_custom_prior)
    m.fit(0, X, y, num_warmup=20, num_samples=20, progress_bar=False, print_summary=False)
    s = m.get_samples()
    assert m.mcmc is not None and s["k_length"].shape == (20,)  # scalar site: no ARD plate in this prior
    assert np.all((s["k_length"] > 0) & (s["k_length"] < 1))    # Uniform(0, 1) support
    mean, draws = m.predict(1, np.linspace(1, 2, 5), n=2)
    assert mean.shape == (5,) and draws.shape == (20, 2, 5)


def test_kernel_prior_with_plate_and_deterministic_scale():
    import gpax_amd

    def prior():
        with gpax_amd.plate("ard", 2):
            length = gpax_amd.sample("k_length", dist.Gamma(2.0, 5.0))
        scale = gpax_amd.deterministic("k_scale", 1.0)
        return {"k_length": length, "k_scale": scale}

    with pytest.warns(UserWarning):
        m = ExactGP(2, "RBF", kernel_prior=prior)
    sites = m._sites()
    assert [(s.name, s.shape) for s in sites] == [("k_length", (2,)), ("noise", ())]
    theta = m._unpack(sites, np.zeros(3))
    assert theta["k_scale"] == 1.0 and theta["k_length"].shape == (2,)


def test_noise_prior_callable_is_accepted_with_the_deprecation_warning():  # gp.py:108-115
    import gpax_amd
    with pytest.warns(FutureWarning):
        m = ExactGP(1, "RBF", noise_prior=lambda: gpax_amd.sample("noise", dist.HalfNormal(0.1)))
    sites = {s.name: s for s in m._sites()}
    assert isinstance(sites["noise"].dist, dist.HalfNormal)


def test_post_processed_prior_draws_are_rejected():
    import gpax_amd

    def prior():
        length = gpax_amd.sample("k_length", dist.LogNormal(0, 1))
        return {"k_length": 2.0 * length, "k_scale": gpax_amd.sample("k_scale", dist.LogNormal(0, 1))}

    with pytest.warns(UserWarning):
        m = ExactGP(1, "RBF", kernel_prior=prior)
    with pytest.raises(NotImplementedError, match="post-processed"):
        m._sites()
    with pytest.raises(RuntimeError):
        gpax_amd.sample("x", dist.Normal())  # outside of a prior callable


def test_mean_fn_prior_callable_and_exact_mean_gradient():
    """gpax/tests/test_gp.py:244-262 (fit with mean_fn + mean_fn_prior); the derivative of the mean function w.r.t.
    its parameters is complex-step (was central differences, O(1e-9) noise): checked to 1e-10 against the
    closed form d/da = x^b, d/db = a x^b log x through the log joint's gradient."""
    X, y = _dummy(12)
    m = ExactGP(1, "RBF", mean_fn=dummy_mean_fn, mean_fn_prior=dummy_mean_fn_priors)
    m.X_train, m.y_train = m._set_data(X, y)
    sites = m._sites()
    names = [s.name for s in sites]
    assert names == ["k_length", "k_scale", "noise", "a", "b"]
    u = np.array([0.1, 0.2, -1.0, 0.3, 2.5])
    val, grad = m._log_joint(sites, u, 1e-6, jacobian=False)
    theta = m._unpack(sites, u)
    x = m.X_train[:, 0]
    p = {"k_length": theta["k_length"], "k_scale": theta["k_scale"], "noise": theta["noise"]}
    yres = y - theta["a"] * x ** theta["b"]
    _, _, _, alpha = ref.exactgp_log_likelihood_grad(m.X_train, y, p, kernel="RBF", yres=yres)
    dm = {"a": x ** theta["b"], "b": theta["a"] * x ** theta["b"] * np.log(x)}
    for name in ("a", "b"):
        i = names.index(name)
        s_ = sites[i]
        xv = np.array([theta[name]])
        expect = (float(alpha @ dm[name]) + s_.dist.grad_log_prob(xv)[0]) * s_.dist.dx_du(u[i:i + 1])[0]
        assert abs(grad[i] - expect) <= 1e-10 * abs(expect), (name, grad[i], expect)
    # a user-supplied derivative takes precedence
    m2 = ExactGP(1, "RBF", mean_fn=dummy_mean_fn, mean_fn_prior=dummy_mean_fn_priors,
                 mean_fn_grad=lambda X_, th: {"a": X_[:, 0] ** th["b"], "b": th["a"] * X_[:, 0] ** th["b"] * np.log(X_[:, 0])})
    m2.X_train, m2.y_train = m.X_train, m.y_train
    _, grad2 = m2._log_joint(sites, u, 1e-6, jacobian=False)
    np.testing.assert_allclose(grad2, grad, rtol=1e-12)
    m.fit(0, X, y, num_warmup=10, num_samples=10, progress_bar=False, print_summary=False)
    assert set(m.get_samples()) == {"k_length", "k_scale", "noise", "a", "b"}


def test_model_returns_the_log_joint():
    """ExactGP.model (gp.py:137-164) = priors + MVN likelihood; evaluated at given params, y = None: priors only."""
    X, y, _, p = bench_inputs.synthetic_problem(30, 2, 4, seed=2)
    m = ExactGP(2, "Matern", noise_prior_dist=dist.HalfNormal(0.5))
    params = {"k_length": np.array([1.0, 1.25]), "k_scale": 1.3, "noise": 0.1}
    lp = dist.LogNormal(0, 1).log_prob(np.array([1.0, 1.25])).sum() + dist.LogNormal(0, 1).log_prob(np.array([1.3]))[0] \
        + dist.HalfNormal(0.5).log_prob(np.array([0.1]))[0]
    assert abs(m.model(X, None, params=params) - lp) < 1e-12
    full = m.model(X, y, params=params)
    assert abs(full - (lp + ref.exactgp_log_likelihood(X, y, p, kernel="Matern"))) < 1e-9
    assert np.isfinite(m.model(X, y))  # default: prior medians
    assert np.isnan(m.model(X, y, params={**params, "k_scale": -1.0}))


def test_sparse_model_returns_the_vfe_log_joint():
    """viSparseGP.model(X, y, Xu) (sparse_gp.py:62-114) = priors + LowRankMVN likelihood - trace_term / 2; it is NOT the
    exact log joint ExactGP.model returns (VERDICT r5 missing #2: the method used to be inherited)."""
    X, y, _, p = bench_inputs.synthetic_problem(30, 2, 4, seed=2)
    Xu = X[::4] + 0.01
    params = {"k_length": np.array([1.0, 1.25]), "k_scale": 1.3, "noise": 0.1}
    m = viSparseGP(2, "Matern", noise_prior_dist=dist.HalfNormal(0.5))
    lp = dist.LogNormal(0, 1).log_prob(np.array([1.0, 1.25])).sum() + dist.LogNormal(0, 1).log_prob(np.array([1.3]))[0] \
        + dist.HalfNormal(0.5).log_prob(np.array([0.1]))[0]
    assert abs(m.model(X, None, params=params) - lp) < 1e-12
    assert abs(m.model(X, None, Xu, params=params) - lp) < 1e-12
    full = m.model(X, y, Xu, params=params)
    assert abs(full - (lp + ref.sparse_bound(X, y, Xu, p, kernel="Matern", jitter=1e-6))) < 1e-9
    exact = ExactGP(2, "Matern", noise_prior_dist=dist.HalfNormal(0.5)).model(X, y, params=params)
    assert full < exact  # a lower bound of the exact log joint at the same theta
    assert abs(m.model(X, y, Xu, params=params, jitter=1e-4) - (lp + ref.sparse_bound(X, y, Xu, p, kernel="Matern", jitter=1e-4))) < 1e-9
    with pytest.raises(ValueError):
        m.model(X, y, params=params)  # no inducing points yet
    m.Xu = Xu
    assert m.model(X, y, params=params) == full
    assert np.isfinite(m.model(X, y))  # default: prior medians
    assert np.isnan(m.model(X, y, Xu, params={**params, "k_scale": -1.0}))
    # after a fit the default Xu is the learnt one, and the value is minus the last loss of the 'delta' guide's objective
    m2 = viSparseGP(2, "Matern")
    m2.fit(get_keys()[0], X, y, inducing_points_ratio=0.3, num_steps=3, progress_bar=False, print_summary=False)
    assert np.isfinite(m2.model(X, y, params=m2.get_samples()))


def test_default_priors_are_overridable_programs_like_the_references():
    """gp.py:222-247: `_sample_noise` / `_sample_kernel_params(output_scale)` ARE the default priors — the reference's own
    subclasses override or re-parametrise them (corgp.py:68 calls _sample_kernel_params(output_scale=False)).  Here they are
    the same programs over gpax_amd.sample / plate / deterministic, traced once: the default sites are unchanged, and a
    subclass that overrides them changes what fit() samples."""
    import gpax_amd
    m = ExactGP(2, "Periodic", lengthscale_prior_dist=dist.Gamma(2.0, 5.0), noise_prior_dist=dist.HalfNormal(0.2))
    assert [(s.name, s.shape, type(s.dist).__name__) for s in m._sites()] == [
        ("k_length", (2,), "Gamma"), ("k_scale", (), "LogNormal"), ("period", (), "LogNormal"), ("noise", (), "HalfNormal")]
    with pytest.raises(RuntimeError):
        m._sample_noise()  # only meaningful while a model traces it

    class UnitScale(ExactGP):
        def _sample_kernel_params(self, output_scale=True):
            return super()._sample_kernel_params(output_scale=False)

        def _sample_noise(self):
            return gpax_amd.sample("noise", dist.HalfNormal(0.3))

    X, y = get_dummy_data()
    u = UnitScale(1, "RBF")
    assert [(s.name, type(s.dist).__name__) for s in u._sites()] == [("k_length", "LogNormal"), ("noise", "HalfNormal")]
    u.fit(get_keys()[0], X, y, num_warmup=10, num_samples=10, progress_bar=False, print_summary=False)
    smp = u.get_samples()
    assert set(smp) == {"k_length", "k_scale", "noise"} and np.all(smp["k_scale"] == 1.0)
    lp = dist.LogNormal(0, 1).log_prob(np.array([0.9]))[0] + dist.HalfNormal(0.3).log_prob(np.array([0.2]))[0]
    assert abs(u.model(X, None, params={"k_length": np.array([0.9]), "noise": 0.2}) - lp) < 1e-12


@pytest.mark.parametrize("kernel", ["RBF", "Matern", "Periodic"])
def test_sample_kernel_and_noise_programs_by_hand(kernel):
    """gpax/tests/test_gp.py:79-127 (test_sample_kernel, test_sample_periodic_kernel, test_sample_noise and the two
    custom-prior tests): the default prior programs run by hand under a seed handler return arrays — draws, which change
    with the prior handed to the constructor."""
    import gpax_amd
    m = ExactGP(1, kernel)
    with gpax_amd.seed(rng_seed=1):
        kernel_params = m._sample_kernel_params()
    period = kernel_params.pop("period")
    assert (period is not None) == (kernel == "Periodic")
    for k, v in kernel_params.items():
        assert k in ("k_length", "k_scale") and isinstance(v, np.ndarray)
    with gpax_amd.seed(rng_seed=1):
        noise1 = m._sample_noise()
    with gpax_amd.seed(rng_seed=1):
        noise2 = ExactGP(1, kernel, noise_prior_dist=dist.HalfNormal(0.1))._sample_noise()
    assert isinstance(noise1, np.ndarray) and not np.array_equal(noise1, noise2)
    with gpax_amd.seed(rng_seed=1):
        l1 = m._sample_kernel_params()["k_length"]
    with gpax_amd.seed(rng_seed=1):
        l2 = ExactGP(1, kernel, lengthscale_prior_dist=dist.Normal(20, 0.1))._sample_kernel_params()["k_length"]
    assert not np.array_equal(l1, l2) and abs(float(l2[0]) - 20) < 1.0


def test_place_prior_helpers_inside_a_mean_fn_prior():
    """priors.py:18-68 with the import swapped: the place_*_prior helpers register their site with the model's trace."""
    from gpax_amd import priors
    X, y = get_dummy_data()

    def mean_fn(x, p):
        return p["a"] * x[:, 0] + p["b"]

    def mean_prior():
        return {"a": priors.place_normal_prior("a", 1.0, 2.0), "b": priors.place_halfnormal_prior("b", 0.5)}

    m = ExactGP(1, "RBF", mean_fn=mean_fn, mean_fn_prior=mean_prior)
    got = {s.name: s.dist for s in m._sites()}
    assert isinstance(got["a"], dist.Normal) and isinstance(got["b"], dist.HalfNormal)
    assert (got["a"].loc, got["a"].scale, got["b"].scale) == (1.0, 2.0, 0.5)
    for fn, args, kind in [(priors.place_lognormal_prior, ("c", 0.0, 0.3), dist.LogNormal),
                           (priors.place_uniform_prior, ("c", None, None, np.array([1.0, 4.0])), dist.Uniform),
                           (priors.place_gamma_prior, ("c", None, None, np.array([1.0, 4.0])), dist.Gamma)]:
        sites = ExactGP(1, "RBF", mean_fn=lambda x, p: p["c"] * x[:, 0], mean_fn_prior=lambda: {"c": fn(*args)})._sites()
        assert isinstance({s.name: s.dist for s in sites}["c"], kind)
    m.fit(get_keys()[0], X, y, num_warmup=5, num_samples=5, progress_bar=False, print_summary=False)
    assert {"a", "b"} <= set(m.get_samples())


def test_sample_from_prior_is_mvn_sample_of_the_prior_draws():
    """gp.py:401-408 against the oracle's mvn_sample: same generator, same consumption order (sites, then eps)."""
    from gpax_amd.utils.utils import rng_from_key
    X = np.linspace(0, 3, 25)[:, None]
    m = ExactGP(1, "RBF")
    out = m.sample_from_prior(5, X, num_samples=3, jitter=1e-5)
    rng = rng_from_key(5)
    for i in range(3):
        theta = {s.name: (s.dist.sample(rng, s.shape) if s.shape else float(s.dist.sample(rng))) for s in m._sites()}
        eps = rng.standard_normal(25)
        K = ref.RBFKernel(X, X, theta, theta["noise"], jitter=1e-5)
        np.testing.assert_allclose(out[i], ref.mvn_sample(np.zeros(25), K, eps[None])[0], rtol=1e-9, atol=1e-12)


def test_nested_plates_follow_numpyros_dimension_rule():
    """without `dim` the OUTER plate is the LAST axis (first free dim counting from -1); explicit dims win."""
    from gpax_amd import dist, plate, sample
    from gpax_amd.infer.primitives import trace_sites

    def auto():
        with plate("a", 3):
            with plate("b", 2):
                return {"x": sample("x", dist.Normal(0.0, 1.0))}

    def explicit():
        with plate("tasks", 3, dim=-2):
            with plate("ard", 2, dim=-1):
                return {"x": sample("x", dist.Normal(0.0, 1.0))}

    def gap():
        with plate("only", 4, dim=-2):
            return {"x": sample("x", dist.Normal(0.0, 1.0))}

    assert trace_sites(auto, "t")[0][0][1] == (2, 3)
    assert trace_sites(explicit, "t")[0][0][1] == (3, 2)
    assert trace_sites(gap, "t")[0][0][1] == (4, 1)
    with pytest.raises(ValueError):
        with plate("p", 2, dim=0):
            pass


def test_vector_valued_mean_function_parameters_get_their_full_jacobian():
    """a plate inside mean_fn_prior makes a site with several entries; d lml / d entry_i = J_i . alpha."""
    from gpax_amd import ExactGP, dist, plate, sample
    from tests.oracle_engine import OracleEngine
    from gpax_amd import _lib
    _lib.set_engine(OracleEngine())

    def mean_fn(x, p):
        return p["w"][0] * x[:, 0] + p["w"][1] * x[:, 0] ** 2 + p["b"]

    def mean_prior():
        with plate("coeffs", 2):
            w = sample("w", dist.Normal(0.0, 2.0))
        return {"w": w, "b": sample("b", dist.Normal(0.0, 1.0))}

    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, (25, 1))
    y = 0.7 * X[:, 0] - 0.4 * X[:, 0] ** 2 + 0.2 + 0.05 * rng.standard_normal(25)
    m = ExactGP(1, "RBF", mean_fn=mean_fn, mean_fn_prior=mean_prior)
    m.X_train, m.y_train = m._set_data(X, y)
    sites = m._sites()
    assert {s.name: tuple(s.shape) for s in sites}["w"] == (2,)
    u = rng.standard_normal(sum(s.size for s in sites)) * 0.3
    val, grad = m._log_joint(sites, u, 1e-6, True)
    num = np.zeros_like(u)
    for i in range(u.size):
        e = np.zeros_like(u)
        e[i] = 1e-6
        num[i] = (m._log_joint(sites, u + e, 1e-6, True)[0] - m._log_joint(sites, u - e, 1e-6, True)[0]) / 2e-6
    np.testing.assert_allclose(grad, num, rtol=2e-6, atol=1e-7)
    J = m._dmean(m.X_train, m._unpack(sites, u), "w")
    np.testing.assert_allclose(J, np.stack([X[:, 0], X[:, 0] ** 2]), rtol=1e-12)
    _lib.set_engine(None)
