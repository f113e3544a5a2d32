"""CPU: pin the oracle.  The reference holds no numeric vectors on this path and cannot be imported
here (no jax), so the restatement is checked against INDEPENDENT computations of the same
quantities and against the committed golden fixtures."""
import os
import subprocess
import sys

import mpmath
import numpy as np
import pytest
import scipy.stats

from oracle import cpu_ref as ref

import bench_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def mp_kernel(name, x, z, ell, scale):
    mpmath.mp.dps = 50
    r2 = sum(((mpmath.mpf(float(a)) - mpmath.mpf(float(b))) / mpmath.mpf(float(l))) ** 2 for a, b, l in zip(x, z, ell))
    if name == "RBF":
        return mpmath.mpf(scale) * mpmath.exp(-r2 / 2)
    r = mpmath.sqrt(r2 + mpmath.mpf("1e-12"))
    s5 = mpmath.sqrt(5) * r
    return mpmath.mpf(scale) * (1 + s5 + mpmath.mpf(5) / 3 * r2) * mpmath.exp(-s5)


@pytest.mark.parametrize("name", ["RBF", "Matern"])
def test_gram_vs_mpmath_direct_formula(name):
    rng = np.random.default_rng(0)
    X, Z = rng.uniform(0, 10, (12, 3)), rng.uniform(0, 10, (9, 3))
    ell = np.array([0.8, 1.5, 2.2])
    K = ref.get_kernel(name)(X, Z, {"k_length": ell, "k_scale": 1.7})
    for i in range(12):
        for j in range(9):
            exact = float(mp_kernel(name, X[i], Z[j], ell, 1.7))
            assert abs(K[i, j] - exact) <= 1e-12 * max(abs(exact), 1e-3)


def test_diagonal_rule_is_shape_equality_not_identity():
    # kernels.py:63: the (noise + jitter) diagonal is added iff X.shape == Z.shape
    rng = np.random.default_rng(1)
    X, Z = rng.uniform(0, 1, (6, 2)), rng.uniform(0, 1, (6, 2))
    p = {"k_length": 1.0, "k_scale": 1.0}
    K0 = ref.RBFKernel(X, Z, p, noise=0.0, jitter=0.0)
    K1 = ref.RBFKernel(X, Z, p, noise=0.5, jitter=1e-6)
    np.testing.assert_allclose(K1 - K0, (0.5 + 1e-6) * np.eye(6), atol=1e-15)
    K2 = ref.RBFKernel(X, Z[:5], p, noise=0.5)
    np.testing.assert_allclose(K2, K0[:, :5], atol=0)


def test_matern_diagonal_uses_sqrt_eps():
    X = np.zeros((3, 2))
    K = ref.MaternKernel(X, X, {"k_length": 1.0, "k_scale": 2.0}, noise=0.0, jitter=0.0)
    r = np.sqrt(1e-12)
    assert np.allclose(K, 2.0 * (1 + np.sqrt(5) * r) * np.exp(-np.sqrt(5) * r), rtol=1e-15)
    assert K[0, 0] < 2.0


@pytest.mark.parametrize("name", ["RBF", "Matern"])
def test_lml_vs_scipy_multivariate_normal(name):
    X, y, _, p = bench_inputs.synthetic_problem(120, 2, 4, seed=3)
    K = ref.get_kernel(name)(X, X, p, p["noise"], jitter=1e-6)
    expect = scipy.stats.multivariate_normal(mean=np.zeros(120), cov=K, allow_singular=False).logpdf(y)
    got = ref.exactgp_log_likelihood(X, y, p, kernel=name)
    assert abs(got - expect) <= 1e-10 * abs(expect)


@pytest.mark.parametrize("name", ["RBF", "Matern"])
def test_gradient_vs_central_differences(name):
    X, y, _, p = bench_inputs.synthetic_problem(80, 2, 4, seed=5)
    g_ell, g_s, g_n, alpha = ref.exactgp_log_likelihood_grad(X, y, p, kernel=name)
    base = np.concatenate([p["k_length"], [p["k_scale"], p["noise"]]])

    def f(t):
        return ref.exactgp_log_likelihood(X, y, {"k_length": t[:2], "k_scale": t[2], "noise": t[3]}, kernel=name)

    fd = np.empty(4)
    for i in range(4):
        h = 1e-6 * base[i]
        a, b = base.copy(), base.copy()
        a[i] += h
        b[i] -= h
        fd[i] = (f(a) - f(b)) / (2 * h)
    np.testing.assert_allclose(np.concatenate([g_ell, [g_s, g_n]]), fd, rtol=2e-7)
    K = ref.get_kernel(name)(X, X, p, p["noise"], jitter=1e-6)
    np.testing.assert_allclose(K @ alpha, y, rtol=1e-9)


@pytest.mark.parametrize("name", ["RBF", "Matern"])
@pytest.mark.parametrize("noiseless", [False, True])
def test_posterior_inverse_route_equals_cholesky_route(name, noiseless):
    X, y, Xn, p = bench_inputs.synthetic_problem(256, 2, 100, seed=7)
    m1, c1 = ref.get_mvn_posterior(X, y, Xn, p, noiseless, kernel=name, route="inv")
    m2, c2 = ref.get_mvn_posterior(X, y, Xn, p, noiseless, kernel=name, route="chol")
    assert np.linalg.norm(m1 - m2) <= 1e-11 * np.linalg.norm(m1)
    assert np.linalg.norm(c1 - c2) <= 1e-11 * np.linalg.norm(c1)


def test_posterior_with_mean_function():
    X, y, Xn, p = bench_inputs.synthetic_problem(60, 1, 25, seed=2)
    p = dict(p, a=0.7)
    mean_fn = lambda x, prm: prm["a"] * x[:, 0]
    m, c = ref.get_mvn_posterior(X, y + 0.7 * X[:, 0], Xn, p, mean_fn=mean_fn, mean_fn_has_params=True)
    m0, c0 = ref.get_mvn_posterior(X, y, Xn, p)
    np.testing.assert_allclose(m, m0 + 0.7 * Xn[:, 0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(c, c0, rtol=0, atol=0)


def test_lowrank_mvn_vs_dense():
    rng = np.random.default_rng(0)
    W, D, y = rng.standard_normal((50, 7)), rng.uniform(0.1, 1, 50), rng.standard_normal(50)
    dense = scipy.stats.multivariate_normal(mean=np.zeros(50), cov=W @ W.T + np.diag(D)).logpdf(y)
    assert abs(ref.lowrank_mvn_log_prob(y, np.zeros(50), W, D) - dense) <= 1e-10 * abs(dense)


def test_sparse_posterior_tends_to_exact_when_inducing_equals_train():
    X, y, Xn, p = bench_inputs.synthetic_problem(40, 1, 15, seed=1)
    m_s, c_s = ref.sparse_posterior(X, y, X.copy(), Xn, p, kernel="RBF", jitter=1e-8)
    m_e, c_e = ref.get_mvn_posterior(X, y, Xn, p, kernel="RBF", jitter=1e-6)
    np.testing.assert_allclose(m_s, m_e, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c_s, c_e, rtol=1e-3, atol=1e-4)


def test_sparse_bound_tends_to_the_exact_log_likelihood_when_inducing_equals_train():
    """The analytic tie that carries the exact leg's pin to the sparse leg (sparse_gp.py:92-114): with
    Xu = X, Qff = Kfu Kuu^-1 Kuf -> Kff, so the trace term -> 0 and LowRankMVN(W W^T + noise I) -> MVN(Kff + noise I): the
    VFE bound becomes the exact log marginal likelihood — and THAT function (exactgp_log_likelihood) is pinned to
    reference-printed output (tests/test_reference_notebook_pins.py).  The bound is a lower bound for every inducing set
    and grows towards the exact value as the set grows towards X."""
    X, y, _, p = bench_inputs.synthetic_problem(60, 1, 3, seed=1)
    order = np.random.default_rng(0).permutation(60)
    for name in ("RBF", "Matern"):
        # Xu = X: the reference's calls as they stand (sparse_gp.py:94-100) give Kuu = Kuf = Kff + jitter I (Kuf is built with
        # the kernel's DEFAULT jitter and X.shape == Xu.shape puts it on the diagonal, kernels.py:63), so Qff = Kff + jitter I
        # exactly, the trace term is -N jitter / noise -> clipped to 0, and the low-rank covariance is Kff + (noise + jitter) I:
        # ExactGP.model's covariance at the same jitter (gp.py:160-164)
        exact = ref.exactgp_log_likelihood(X, y, p, kernel=name, jitter=1e-6)
        full = ref.sparse_bound(X, y, X.copy(), p, kernel=name, jitter=1e-6)
        assert abs(full - exact) <= 1e-7 * abs(exact), (name, full, exact)
        # nested inducing sets: a lower bound of the exact value (no jitter on Kff there: sparse_gp.py:100) that grows with
        # the set (Titsias 2009); jitter 1e-6 on Kuu loosens both statements by O(m jitter / noise)
        exact0 = ref.exactgp_log_likelihood(X, y, p, kernel=name, jitter=0.0)
        prev = -np.inf
        for m in (6, 12, 24, 48):
            b = ref.sparse_bound(X, y, X[order[:m]], p, kernel=name, jitter=1e-6)
            assert prev - 1e-3 <= b <= exact0 + 1e-3, (name, m, b, prev, exact0)
            prev = b
        assert prev > exact0 - 0.05 * abs(exact0)


def test_golden_fixtures_match_oracle():
    g = load("gram")
    for c in range(int(g["ncases"])):
        kind, scale, noise, jitter = g[f"c{c}_meta"]
        name = "RBF" if kind == 0 else "Matern"
        ell = g[f"c{c}_ell"]
        K = ref.get_kernel(name)(g[f"c{c}_X"], g[f"c{c}_Z"], {"k_length": ell, "k_scale": scale}, noise=noise,
                                 jitter=jitter)
        np.testing.assert_array_equal(K, g[f"c{c}_K"])
    p = load("posterior")
    for c in range(0, int(p["ncases"]), 5):
        kind, noiseless, jitter = p[f"c{c}_meta"]
        th = p[f"c{c}_theta"]
        prm = {"k_length": th[:-2], "k_scale": th[-2], "noise": th[-1]}
        m, cv = ref.get_mvn_posterior(p[f"c{c}_X"], p[f"c{c}_y"], p[f"c{c}_Xn"], prm, bool(noiseless),
                                      kernel="RBF" if kind == 0 else "Matern", jitter=jitter)
        np.testing.assert_allclose(m, p[f"c{c}_mean"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(cv, p[f"c{c}_cov"], rtol=1e-10, atol=1e-12)


def test_utils_restatement():
    u = load("utils")
    a, b, c = ref.preprocess_sparse_image(u["img34"])
    np.testing.assert_array_equal(a, [[0, 1], [1, 0], [2, 3]])
    np.testing.assert_array_equal(b, [1.5, 0.25, -2.0])
    assert c.shape == (12, 2) and tuple(c[1]) == (0, 1) and tuple(c[4]) == (1, 0)  # row-major order
    parts = ref.split_in_batches(np.arange(10), 4)
    assert [len(q) for q in parts] == [4, 4, 2]
    with pytest.raises(UnboundLocalError):
        ref.split_in_batches(np.arange(3), 4)  # the reference's bug (utils.py:48), restated
    chunks = ref.split_dict({"a": np.arange(10), "b": np.arange(20).reshape(10, 2)}, 4)
    assert [len(q["a"]) for q in chunks] == [4, 4, 2] and chunks[1]["b"].shape == (4, 2)


def test_full_pass_equals_the_separate_restatements():
    """exactgp_full_pass (one factorisation; used by the full-size GPU parity tests) is the arithmetic of
    exactgp_log_likelihood + get_mvn_posterior(route='chol') + mvn_sample, bit for bit."""
    X, y, Xn, p = bench_inputs.synthetic_problem(150, 2, 40, seed=11)
    eps = np.random.default_rng(0).standard_normal((3, 40))
    for name in ("RBF", "Matern"):
        lml, mean, cov, draws, alpha = ref.exactgp_full_pass(X, y, Xn, p, eps, False, kernel=name)
        assert lml == ref.exactgp_log_likelihood(X, y, p, kernel=name)
        m2, c2 = ref.get_mvn_posterior(X, y, Xn, p, False, kernel=name, route="chol")
        np.testing.assert_array_equal(mean, m2)
        np.testing.assert_array_equal(cov, c2)
        np.testing.assert_array_equal(draws, ref.mvn_sample(m2, c2, eps))
        K = ref.get_kernel(name)(X, X, p, p["noise"])
        np.testing.assert_allclose(K @ alpha, y, rtol=1e-9, atol=1e-9)


def test_blocked_gradient_restatement_equals_the_unblocked_one():
    """oracle.exactgp_log_likelihood_grad_blocked (row blocks, K^-1 from the Cholesky factor: what the GPU test of the
    headline size N = 16384 runs on the host) against exactgp_log_likelihood_grad (the faithful, memory-hungry form)."""
    from bench_inputs import synthetic_problem
    for name, N, block in (("RBF", 300, 128), ("Matern", 515, 100), ("Matern", 64, 1024)):
        X, y, _, p = synthetic_problem(N, 2, 4, seed=N)
        a = ref.exactgp_log_likelihood_grad(X, y, p, kernel=name)
        b = ref.exactgp_log_likelihood_grad_blocked(X, y, p, kernel=name, block=block)
        sc = max(np.abs(a[0]).max(), abs(a[1]), abs(a[2]))
        np.testing.assert_allclose(b[0], a[0], rtol=0, atol=1e-11 * sc)
        assert abs(b[1] - a[1]) <= 1e-11 * sc and abs(b[2] - a[2]) <= 1e-11 * sc
        np.testing.assert_allclose(b[3], a[3], rtol=1e-10, atol=1e-12 * np.abs(a[3]).max())
