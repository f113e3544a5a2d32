"""GPU: DIRECT HIP-vs-oracle parity at BASELINE.json's sizes — the oracle
(oracle/cpu_ref.py, restating gpax/models/gp.py:137-164,253-293,351-399) run on the host beside the device path:

    C2  RBF     N =  4096, d = 2, M = 1024   lml, gradient, posterior (reference's explicit-inverse route), draws
    C4  RBF     N =  8192, d = 3, M = 1024   one theta as C2; the S = 1000 sweep once (determinism + 4 spot samples)
    C3  Matern  N = 16384, d = 2, M = 1024   lml, alpha, posterior, draws (one host factorisation, ~1-2 min)
    C5  Matern  512 x 512 image, N ~ 16384   exact viGP: SVI steps + predict_in_batches over all 262 144 pixels
    C5  Matern  same image, M_ind = 2039     viSparseGP: VFE bound, its gradient (central differences of the oracle),
                                             Woodbury posterior on a pixel subset, predict over all 262 144 pixels

Tolerances are SURVEY.md §8c's: |d lml| <= 1e-10 |lml|, |d mean| <= 1e-8 |mean|, |d cov|_F <= 1e-8 |k_pp|_F,
draws 1e-8 — two orders inside the north-star bar (1e-6)."""
import numpy as np
import pytest

from bench_inputs import synthetic_problem, synthetic_sparse_image, synthetic_theta_samples
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu

JIT = 1e-6


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def check_one_theta(engine, kind, name, X, y, Xn, p, eps, route, want_grad):
    engine.set_train(X)
    lml, info = engine.factor(kind, p["k_length"], p["k_scale"], p["noise"], JIT, y)
    assert info == 0
    mean, cov, var = engine.posterior(Xn, p["noise"], JIT, want_cov=True, want_var=True)
    draws, dinfo = engine.mvn_draw(eps)
    assert dinfo == 0
    kpp = ref.get_kernel(name)(Xn, Xn, p, p["noise"], jitter=JIT)
    if route == "inv":  # the reference's own route (gp.py:271-273)
        e_lml = ref.exactgp_log_likelihood(X, y, p, kernel=name, jitter=JIT)
        e_mean, e_cov = ref.get_mvn_posterior(X, y, Xn, p, False, kernel=name, jitter=JIT, route="inv")
        e_draws = ref.mvn_sample(e_mean, e_cov, eps)
        e_alpha = None
    else:
        e_lml, e_mean, e_cov, e_draws, e_alpha = ref.exactgp_full_pass(X, y, Xn, p, eps, False, kernel=name, jitter=JIT)
    assert abs(lml - e_lml) <= 1e-10 * abs(e_lml), (lml, e_lml)
    assert relerr(mean, e_mean) < 1e-8
    assert np.linalg.norm(cov - e_cov) / np.linalg.norm(kpp) < 1e-8
    assert np.linalg.norm(var - np.diag(e_cov)) / np.linalg.norm(np.diag(kpp)) < 1e-8
    assert relerr(draws, e_draws) < 1e-8
    if want_grad or e_alpha is not None:
        engine.factor(kind, p["k_length"], p["k_scale"], p["noise"], JIT, y)
        g_ell, g_scale, g_noise, alpha = engine.lml_grad()
        if want_grad:
            # (the blocked restatement: same formula, N^2 memory — tests/test_oracle.py holds it against the plain one)
            e_ell, e_scale, e_noise, e_alpha = ref.exactgp_log_likelihood_grad_blocked(X, y, p, kernel=name, jitter=JIT)
            sc = max(np.abs(e_ell).max(), abs(e_scale), abs(e_noise))
            np.testing.assert_allclose(g_ell, e_ell, rtol=1e-8, atol=1e-8 * sc)
            assert abs(g_scale - e_scale) <= 1e-8 * sc and abs(g_noise - e_noise) <= 1e-8 * sc
        assert relerr(alpha, e_alpha) < 1e-8


def test_c2_rbf_n4096_vs_oracle(engine):
    N, d, M = 4096, 2, 1024
    X, y, Xn, p = synthetic_problem(N, d, M, seed=0)
    eps = np.random.default_rng(2).standard_normal((2, M))
    check_one_theta(engine, 0, "RBF", X, y, Xn, p, eps, route="inv", want_grad=True)


def test_c4_shape_n8192_d3_one_theta_vs_oracle(engine):
    N, d, M = 8192, 3, 1024
    X, y, Xn, p = synthetic_problem(N, d, M, seed=0)
    eps = np.random.default_rng(2).standard_normal((1, M))
    check_one_theta(engine, 0, "RBF", X, y, Xn, p, eps, route="inv", want_grad=True)


def test_c4_sweep_s1000_n8192_d3(engine):
    """BASELINE.json configs[3] on one GPU: the full 1000-sample sweep once (contexts in flight as predict() runs
    it), bit-reproducible, 2 spot samples against the oracle, y_means.mean(0) as gp.py:399."""
    from gpax_amd import _lib

    N, d, M, S = 8192, 3, 1024, 1000
    X, y, Xn, _ = synthetic_problem(N, d, M, seed=0)
    th = synthetic_theta_samples(S, d, seed=1)
    eps = np.random.default_rng(2).standard_normal((S, 1, M))
    engines = _lib.get_sweep_engines(0)
    m1, d1, i1 = _lib.concurrent_sweep(engines, X, 0, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, JIT, eps)
    assert m1.shape == (S, M) and d1.shape == (S, 1, M) and np.all(i1 == 0)
    assert np.isfinite(m1).all() and np.isfinite(d1).all()
    # the same table through ONE context (other batch boundaries, no threads): identical sample by sample
    engine.set_train(X)
    sub = np.r_[0:40, 960:1000]
    m2, d2, i2 = engine.predict_sweep(0, th["k_length"][sub], th["k_scale"][sub], th["noise"][sub], y, Xn, False, JIT,
                                      eps[sub])
    np.testing.assert_array_equal(m1[sub], m2)
    np.testing.assert_array_equal(d1[sub], d2)
    for s in (333, 999):  # ~7 s of CPU oracle each
        q = {k: v[s] for k, v in th.items()}
        e_mean, e_draw = ref.predict_one(X, y, Xn, q, eps[s], False, kernel="RBF", jitter=JIT, route="chol")
        assert relerr(m1[s], e_mean) < 1e-8
        assert relerr(d1[s], e_draw) < 1e-8
    assert relerr(m1.mean(0), m1.astype(np.longdouble).mean(0).astype(np.float64)) < 1e-14


def test_c3_matern_n16384_vs_oracle(engine):
    N, d, M = 16384, 2, 1024
    X, y, Xn, p = synthetic_problem(N, d, M, seed=0)
    eps = np.random.default_rng(2).standard_normal((1, M))
    check_one_theta(engine, 1, "Matern", X, y, Xn, p, eps, route="chol", want_grad=False)


def test_c3_gradient_n16384_vs_oracle_and_tree_vs_sweep(monkeypatch):
    """the fit step's gradient at the HEADLINE size.  `linv_t_tree` runs 7 levels deep at N = 16384
    and K^-1 = W W^T has 129 tile rows there; until round 4 both were oracle-checked to N = 8192 only.  gpx_lml_grad (tree
    path) against the oracle's analytic gradient (gpax/models/gp.py:137-164 gets it from JAX autodiff) at 1e-8, and
    against the right-looking sweep (GPX_LINVT=sweep, another summation order) at 1e-10."""
    from gpax_amd import _lib

    N, d = 16384, 2
    X, y, _, p = synthetic_problem(N, d, 8, seed=0)
    got = {}
    for mode in ("tree", "sweep"):
        monkeypatch.setenv("GPX_LINVT", mode)
        e = _lib.Engine(0)
        e.set_train(X)
        lml, info = e.factor(1, p["k_length"], p["k_scale"], p["noise"], JIT, y)
        assert info == 0
        got[mode] = (lml,) + tuple(e.lml_grad())
        e.close()
    monkeypatch.delenv("GPX_LINVT")
    e_ell, e_scale, e_noise, e_alpha = ref.exactgp_log_likelihood_grad_blocked(X, y, p, kernel="Matern", jitter=JIT)
    sc = max(np.abs(e_ell).max(), abs(e_scale), abs(e_noise))
    for mode, tol in (("tree", 1e-8), ("sweep", 1e-8)):
        lml, g_ell, g_scale, g_noise, alpha = got[mode]
        np.testing.assert_allclose(g_ell, e_ell, rtol=0, atol=tol * sc)
        assert abs(g_scale - e_scale) <= tol * sc and abs(g_noise - e_noise) <= tol * sc
        assert relerr(alpha, e_alpha) < 1e-8
    t, s_ = got["tree"], got["sweep"]
    assert t[0] == s_[0]  # the factorisation is the same launch sequence
    np.testing.assert_allclose(t[1], s_[1], rtol=0, atol=1e-10 * sc)
    assert abs(t[2] - s_[2]) <= 1e-10 * sc and abs(t[3] - s_[3]) <= 1e-10 * sc
    assert relerr(t[4], s_[4]) < 1e-10


def test_c5_exact_vigp_on_the_512x512_image():
    """BASELINE.json configs[4], the exact viGP leg (gpax/models/vigp.py:77-185): N ~ 16384 training pixels, a few
    SVI steps (each = one device fit step), predict_in_batches over the 262 144 pixels in slices of 1000 with ONE
    factorisation, compared with the oracle's vigp_predict on a pixel subset."""
    from gpax_amd import _lib, viGP
    from gpax_amd.utils import preprocess_sparse_image

    _lib.set_engine(None)
    img, sparse = synthetic_sparse_image(512, 512, 0.0625, seed=3)
    X, y, X_full = preprocess_sparse_image(sparse)
    assert 15000 < X.shape[0] < 18000 and X_full.shape == (512 * 512, 2)
    m = viGP(2, "Matern", lengthscale_prior_dist=None)
    m.fit(0, X, y, num_steps=6, step_size=5e-2, progress_bar=False, print_summary=False)
    assert len(m.loss) == 6 and np.all(np.isfinite(m.loss)) and m.loss[-1] < m.loss[0]
    theta = m.get_samples()
    assert set(theta) == {"k_length", "k_scale", "noise"}
    # a well-conditioned theta for the reconstruction (6 SVI steps from the prior median are not a converged fit)
    theta = {"k_length": np.array([25.0, 25.0]), "k_scale": np.float64(1.0), "noise": np.float64(1e-2)}
    mean, var = m.predict_in_batches(0, X_full, batch_size=1000, samples=theta, noiseless=True)
    assert mean.shape == (512 * 512,) and var.shape == (512 * 512,)
    assert np.isfinite(mean).all() and np.all(var > 0)
    rmse = np.sqrt(np.mean((mean.reshape(512, 512) - img) ** 2))
    assert rmse < 0.05, rmse
    # slices of one factorisation == predict() on the slice alone (vigp.py:129-151 re-inverts per slice)
    sl = slice(100_000, 101_000)
    m1, v1 = m.predict(0, X_full[sl], samples=theta, noiseless=True)
    np.testing.assert_array_equal(mean[sl], m1)
    np.testing.assert_array_equal(var[sl], v1)
    # oracle: viGP.predict = (mean, diag cov) of get_mvn_posterior at the guide median (vigp.py:184-185)
    idx = np.random.default_rng(5).choice(512 * 512, 256, replace=False)
    p = {"k_length": theta["k_length"], "k_scale": float(theta["k_scale"]), "noise": float(theta["noise"])}
    e_mean, e_var = ref.vigp_predict(X, y, X_full[idx], p, noiseless=True, kernel="Matern", jitter=JIT, route="chol")
    assert relerr(mean[idx], e_mean) < 1e-8
    assert np.linalg.norm(var[idx] - e_var) / np.linalg.norm(e_var) < 1e-6  # k_ss - |V|^2 cancels ~1e3 : 1 here


def _c5_sparse_problem():
    from gpax_amd.utils import get_keys, initialize_inducing_points, preprocess_sparse_image
    img, sparse = synthetic_sparse_image(512, 512, 0.0625, seed=3)
    X, y, X_full = preprocess_sparse_image(sparse)
    Xu = initialize_inducing_points(X, 0.125, "random", get_keys(0)[0])  # sparse_gp.py:151-154: ratio 0.125 -> M_ind = 2039
    assert 15000 < X.shape[0] < 18000 and Xu.shape == (int(X.shape[0] * 0.125), 2) and Xu.shape[0] > 2000
    return img, X, y - y.mean(), y.mean(), X_full, Xu


def test_c5_visparsegp_bound_and_posterior_at_m2048_vs_oracle(engine):
    """BASELINE.json configs[4], the viSparseGP leg at ITS size: the VFE bound of
    gpax/models/sparse_gp.py:62-114 and the Woodbury posterior of sparse_gp.py:173-223 on the 512 x 512 image
    (N = 16 316 measured pixels, 'random' inducing points at ratio 0.125 -> M_ind = 2039, Matern), device against
    oracle.  Conditioning: Kuu carries only the 1e-6 jitter; at this theta cond(Kuu) = 5.7e6 and a 0.1 % change of
    the jitter moves the oracle's own bound by 3e-7 relative, so 1e-9 on the bound and 1e-7 on the posterior is what
    fp64 can promise here (the north-star bar is 1e-6)."""
    img, X, y, ybar, X_full, Xu = _c5_sparse_problem()
    p = {"k_length": np.array([25.0, 25.0]), "k_scale": 1.0, "noise": 1e-2}
    engine.set_train(X)
    bound, info, _ = engine.sgp_bound(1, p["k_length"], p["k_scale"], p["noise"], JIT, Xu, y, want_grad=False)
    assert info == 0
    e_bound = ref.sparse_bound(X, y, Xu, p, kernel="Matern", jitter=JIT)
    print(f"C5 sparse bound: device {bound:.9f} oracle {e_bound:.9f} rel {abs(bound - e_bound) / abs(e_bound):.2e}")
    assert abs(bound - e_bound) <= 1e-9 * abs(e_bound)
    # the bound never exceeds the exact log marginal likelihood of the same theta (device, exact GP)
    lml, info2 = engine.factor(1, p["k_length"], p["k_scale"], p["noise"], JIT, y)
    assert info2 == 0 and bound <= lml
    # Woodbury posterior on a pixel subset, with and without observation noise on the prediction
    idx = np.random.default_rng(5).choice(512 * 512, 300, replace=False)
    for noiseless in (False, True):
        noise_p = 0.0 if noiseless else p["noise"]
        mean, cov, var, info3 = engine.sgp_posterior(1, p["k_length"], p["k_scale"], p["noise"], JIT, Xu, y, X_full[idx],
                                                     noise_p, want_cov=True, want_var=True)
        assert info3 == 0
        e_mean, e_cov = ref.sparse_posterior(X, y, Xu, X_full[idx], p, noiseless, kernel="Matern", jitter=JIT)
        print(f"C5 sparse posterior (noiseless={noiseless}): mean rel {relerr(mean, e_mean):.2e}, "
              f"cov max abs {np.abs(cov - e_cov).max():.2e}, var max abs {np.abs(var - np.diag(e_cov)).max():.2e}")
        assert relerr(mean, e_mean) <= 1e-7
        assert np.abs(cov - e_cov).max() <= 1e-7 * p["k_scale"]
        assert np.abs(var - np.diag(e_cov)).max() <= 1e-7 * p["k_scale"]
    # the whole image through the model API (predict_in_batches -> device-sized slices), against the subset above
    from gpax_amd import _lib, viSparseGP
    _lib.set_engine(None)
    m = viSparseGP(2, "Matern")
    m.X_train, m.y_train = m._set_data(X, y)
    m._data_version += 1
    m.Xu = Xu
    theta = {"k_length": p["k_length"], "k_scale": np.float64(p["k_scale"]), "noise": np.float64(p["noise"])}
    mean_all, var_all = m.predict_in_batches(0, X_full, batch_size=1000, samples=theta, noiseless=True)
    assert mean_all.shape == (512 * 512,) and np.isfinite(mean_all).all() and np.all(var_all > 0)
    assert relerr(mean_all[idx], e_mean) <= 1e-7  # e_mean: the noiseless pass, computed last above
    rmse = np.sqrt(np.mean((mean_all.reshape(512, 512) + ybar - img) ** 2))
    assert rmse < 0.05, rmse


def test_c5_visparsegp_gradient_at_m2048_vs_central_differences_of_the_oracle(engine):
    """d bound / d (k_length[0], noise, one inducing coordinate) at C5's size against central differences of the
    oracle's sparse_bound (six host evaluations).  jitter = 1e-4 keeps cond(Kuu) ~ 6e4 so that a difference quotient
    of the oracle carries ~7 digits (the same choice as the small-size gradient test, test_gpu_sparse.py)."""
    img, X, y, ybar, X_full, Xu = _c5_sparse_problem()
    p = {"k_length": np.array([25.0, 25.0]), "k_scale": 1.0, "noise": 1e-2}
    jit = 1e-4
    engine.set_train(X)
    bound, info, g = engine.sgp_bound(1, p["k_length"], p["k_scale"], p["noise"], jit, Xu, y)
    assert info == 0 and np.isfinite(g["Xu"]).all() and np.isfinite(g["yres"]).all()

    def f(ell, noise, xu):
        return ref.sparse_bound(X, y, xu, {"k_length": ell, "k_scale": 1.0, "noise": noise}, kernel="Matern", jitter=jit)

    ell0 = p["k_length"]
    h = 1e-4 * ell0[0]
    fd = (f(ell0 + np.array([h, 0.0]), 1e-2, Xu) - f(ell0 - np.array([h, 0.0]), 1e-2, Xu)) / (2 * h)
    print(f"C5 sparse d/d ell0: device {g['k_length'][0]:.6e} fd {fd:.6e}")
    assert abs(fd - g["k_length"][0]) <= 1e-4 * max(abs(g["k_length"][0]), abs(g["k_length"][1]))
    h = 1e-6
    fd = (f(ell0, 1e-2 + h, Xu) - f(ell0, 1e-2 - h, Xu)) / (2 * h)
    print(f"C5 sparse d/d noise: device {g['noise']:.6e} fd {fd:.6e}")
    assert abs(fd - g["noise"]) <= 1e-4 * abs(g["noise"])
    a_ = int(np.argmax(np.abs(g["Xu"][:, 0])))  # the inducing point the bound is most sensitive to
    h = 1e-3
    xp, xm = Xu.copy(), Xu.copy()
    xp[a_, 0] += h
    xm[a_, 0] -= h
    fd = (f(ell0, 1e-2, xp) - f(ell0, 1e-2, xm)) / (2 * h)
    print(f"C5 sparse d/d Xu[{a_},0]: device {g['Xu'][a_, 0]:.6e} fd {fd:.6e}")
    assert abs(fd - g["Xu"][a_, 0]) <= 1e-3 * np.abs(g["Xu"]).max()
