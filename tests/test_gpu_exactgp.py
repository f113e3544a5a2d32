"""GPU parity of the exact-GP pipeline (lml, gradient, posterior, draw, sweep) vs the oracle."""
import numpy as np
import pytest

from oracle import cpu_ref as ref

import bench_inputs

pytestmark = pytest.mark.gpu

KINDS = [(0, "RBF"), (1, "Matern")]


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("kind,name", KINDS)
@pytest.mark.parametrize("N,d", [(8, 1), (64, 1), (127, 2), (128, 2), (300, 3), (1100, 2)])
def test_lml_matches_oracle(engine, kind, name, N, d):
    X, y, _, params = bench_inputs.synthetic_problem(N, d, 4, seed=N + d)
    engine.set_train(X)
    lml, info = engine.factor(kind, params["k_length"], params["k_scale"], params["noise"], 1e-6, y)
    assert info == 0
    expect = ref.exactgp_log_likelihood(X, y, params, kernel=name, jitter=1e-6)
    assert abs(lml - expect) <= 1e-10 * abs(expect)


@pytest.mark.parametrize("kind,name", KINDS)
@pytest.mark.parametrize("N,d", [(50, 1), (200, 2), (391, 3)])
def test_lml_grad_matches_oracle(engine, kind, name, N, d):
    X, y, _, params = bench_inputs.synthetic_problem(N, d, 4, seed=3 * N + d)
    engine.set_train(X)
    lml, info = engine.factor(kind, params["k_length"], params["k_scale"], params["noise"], 1e-6, y)
    g_ell, g_scale, g_noise, alpha = engine.lml_grad()
    e_ell, e_scale, e_noise, e_alpha = ref.exactgp_log_likelihood_grad(X, y, params, kernel=name, jitter=1e-6)
    scale = max(np.abs(e_ell).max(), abs(e_scale), abs(e_noise))
    np.testing.assert_allclose(g_ell, e_ell, rtol=1e-8, atol=1e-8 * scale)
    assert abs(g_scale - e_scale) <= 1e-8 * scale
    assert abs(g_noise - e_noise) <= 1e-8 * scale
    assert relerr(alpha, e_alpha) < 1e-9


@pytest.mark.parametrize("kind,name", KINDS)
@pytest.mark.parametrize("N,d,M", [(8, 1, 5), (100, 2, 33), (256, 2, 128), (700, 3, 260)])
@pytest.mark.parametrize("noiseless", [False, True])
def test_posterior_matches_oracle_inverse_route(engine, kind, name, N, d, M, noiseless):
    X, y, Xnew, params = bench_inputs.synthetic_problem(N, d, M, seed=N + M)
    engine.set_train(X)
    engine.factor(kind, params["k_length"], params["k_scale"], params["noise"], 1e-6, y)
    noise_p = 0.0 if noiseless else params["noise"]
    mean, cov, var = engine.posterior(Xnew, noise_p, 1e-6, want_cov=True, want_var=True)
    m_ref, c_ref = ref.get_mvn_posterior(X, y, Xnew, params, noiseless, kernel=name, jitter=1e-6, route="inv")
    kpp = ref.get_kernel(name)(Xnew, Xnew, params, noise_p, jitter=1e-6)
    # SURVEY §8c tolerances: |dmean|/|mean| <= 1e-8, |dcov|_F/|k_pp|_F <= 1e-8 (cond(K) <= 1e6)
    assert relerr(mean, m_ref) < 1e-8
    assert np.linalg.norm(cov - c_ref) / np.linalg.norm(kpp) < 1e-8
    assert np.linalg.norm(var - np.diag(c_ref)) / np.linalg.norm(np.diag(kpp)) < 1e-8
    np.testing.assert_array_equal(cov, cov.T)


@pytest.mark.parametrize("kind,name", KINDS)
def test_draw_given_eps_matches_oracle(engine, kind, name):
    N, d, M, n = 150, 2, 70, 5
    X, y, Xnew, params = bench_inputs.synthetic_problem(N, d, M, seed=11)
    eps = np.random.default_rng(2).standard_normal((n, M))
    engine.set_train(X)
    engine.factor(kind, params["k_length"], params["k_scale"], params["noise"], 1e-6, y)
    mean, cov, _ = engine.posterior(Xnew, params["noise"], 1e-6)
    draws, info = engine.mvn_draw(eps)
    assert info == 0
    m_ref, y_ref = ref.predict_one(X, y, Xnew, params, eps, False, kernel=name, jitter=1e-6, route="chol")
    assert relerr(draws, y_ref) < 1e-8


def test_noiseless_invariants(engine):
    # gpax/tests/test_gp.py:155-170: same call twice bit-identical; mean(noiseless) == mean(noisy)
    X, y, Xnew, params = bench_inputs.synthetic_problem(90, 1, 40, seed=4)
    engine.set_train(X)
    engine.factor(0, params["k_length"], params["k_scale"], params["noise"], 1e-6, y)
    m1, c1, _ = engine.posterior(Xnew, params["noise"], 1e-6)
    m2, c2, _ = engine.posterior(Xnew, params["noise"], 1e-6)
    m3, c3, _ = engine.posterior(Xnew, 0.0, 1e-6)
    np.testing.assert_array_equal(m1, m2)
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(m1, m3)
    assert not np.allclose(c1, c3)


def test_non_pd_theta_gives_nan_not_crash(engine):
    # gpax/tests/test_gp.py:196-206 feeds N(0,1) "samples": negative variances must not crash
    X, y, Xnew, params = bench_inputs.synthetic_problem(60, 1, 20, seed=8)
    engine.set_train(X)
    lml, info = engine.factor(0, [1.0], -0.7, 0.1, 1e-6, y)
    assert info > 0 and np.isnan(lml)


@pytest.mark.parametrize("kind,name", KINDS)
def test_predict_sweep_matches_oracle(engine, kind, name):
    N, d, M, S, n = 200, 3, 90, 6, 2
    X, y, Xnew, _ = bench_inputs.synthetic_problem(N, d, M, seed=21)
    samples = bench_inputs.synthetic_theta_samples(S, d, seed=1)
    eps = np.random.default_rng(2).standard_normal((S, n, M))
    engine.set_train(X)
    means, draws, infos = engine.predict_sweep(kind, samples["k_length"], samples["k_scale"], samples["noise"], y,
                                               Xnew, False, 1e-6, eps)
    assert np.all(infos == 0)
    mm, yy, all_means = ref.predict(X, y, Xnew, samples, eps, False, kernel=name, jitter=1e-6, route="chol")
    assert relerr(means, all_means) < 1e-8
    assert relerr(draws, yy) < 1e-8


def test_predict_sweep_bad_sample_is_nan_filled(engine):
    N, d, M, S, n = 80, 1, 30, 3, 1
    X, y, Xnew, _ = bench_inputs.synthetic_problem(N, d, M, seed=5)
    samples = bench_inputs.synthetic_theta_samples(S, d, seed=1)
    samples["k_scale"][1] = -1.0
    eps = np.random.default_rng(2).standard_normal((S, n, M))
    engine.set_train(X)
    means, draws, infos = engine.predict_sweep(0, samples["k_length"], samples["k_scale"], samples["noise"], y,
                                               Xnew, False, 1e-6, eps)
    assert infos[0] == 0 and infos[2] == 0 and infos[1] != 0
    assert np.isnan(draws[1]).all() and np.isfinite(draws[0]).all() and np.isfinite(draws[2]).all()


def test_full_size_roundtrip_properties(engine):
    """C2-size (N=4096) size-independent checks: K alpha = y through the factor, and the
    posterior at the training points reproduces the closed form mean = K_f (K_f + s I)^-1 y."""
    N, d = 4096, 2
    X, y, _, params = bench_inputs.synthetic_problem(N, d, 4, seed=0)
    engine.set_train(X)
    lml, info = engine.factor(0, params["k_length"], params["k_scale"], params["noise"], 1e-6, y)
    assert info == 0 and np.isfinite(lml)
    sub = np.arange(0, N, 16)
    mean, cov, var = engine.posterior(X[sub], params["noise"], 1e-6, want_cov=True, want_var=True)
    # mean at training inputs = y - (noise + jitter) * alpha  =>  alpha recoverable; check K alpha = y
    lml2, _ = engine.factor(0, params["k_length"], params["k_scale"], params["noise"], 1e-6, y)
    assert lml2 == lml  # determinism
    g_ell, g_scale, g_noise, alpha = engine.lml_grad()
    resid = y[sub] - (params["noise"] + 1e-6) * alpha[sub]
    assert relerr(mean, resid) < 1e-8
    K_rows = ref.RBFKernel(X[sub], X, params)  # (len(sub), N) cross block, no diagonal term
    K_rows[np.arange(len(sub)), sub] += params["noise"] + 1e-6
    assert relerr(K_rows @ alpha, y[sub]) < 1e-8
    assert np.all(var > 0) and np.allclose(var, np.diag(cov), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("kind,name", KINDS)
@pytest.mark.parametrize("strided", [False, True])
def test_batched_sweep_is_independent_of_batch_size(engine, kind, name, strided, monkeypatch):
    # the vmap over samples runs as a grid dimension (B samples per launch): every sample's arithmetic is the
    # single-sample arithmetic, so results must be BIT-identical for any B, including ragged last batches
    N, d, M, S, n = 300, 2, 70, 11, 2
    X, y, Xnew, params = bench_inputs.synthetic_problem(N, d, M, seed=21)
    th = bench_inputs.synthetic_theta_samples(S, d, seed=22)
    rng = np.random.default_rng(23)
    eps = rng.standard_normal((S, n, M))
    yres = y[None, :] + 0.01 * rng.standard_normal((S, N)) if strided else y
    th["noise"][4] = -5.0  # a non-PD sample in the middle of a batch must only poison itself
    engine.set_train(X)
    outs = {}
    for B in ("1", "4", "11", "0"):
        monkeypatch.setenv("GPX_SWEEP_BATCH", B)
        before = engine.sweep_stats()
        outs[B] = engine.predict_sweep(kind, th["k_length"], th["k_scale"], th["noise"], yres, Xnew, False, 1e-6, eps)
        nb, ns, last = engine.sweep_stats()
        assert ns - before[1] == S
        assert last == (int(B) if B != "0" else S)  # auto: 4 (16384/Np)^2 >> S at this size
        assert nb - before[0] == -(-S // last)
    for B in ("4", "11", "0"):
        for a, b in zip(outs["1"], outs[B]):
            np.testing.assert_array_equal(a, b)
    means, draws, infos = outs["0"]
    assert infos[4] != 0 and np.all(np.isnan(means[4])) and np.all(np.isnan(draws[4]))
    ok = [s for s in range(S) if s != 4]
    assert np.all(infos[ok] == 0)
    for s in ok[:4]:
        p = {"k_length": th["k_length"][s], "k_scale": th["k_scale"][s], "noise": th["noise"][s]}
        ys = yres[s] if strided else y
        m_ref, c_ref = ref.get_mvn_posterior(X, ys, Xnew, p, False, kernel=name, jitter=1e-6, route="inv")
        assert relerr(means[s], m_ref) < 1e-8
        assert relerr(draws[s], ref.mvn_sample(m_ref, c_ref, eps[s])) < 1e-6


@pytest.mark.parametrize("kind,name", KINDS + [(2, "Periodic")])
@pytest.mark.parametrize("N,d", [(90, 1), (391, 3), (700, 2)])
def test_fit_batch_equals_single_fit_steps(engine, kind, name, N, d):
    # gpx_fit_batch = B x (gpx_factor + gpx_lml_grad) with the chain as a grid dimension: bit-identical entries
    B = 5
    X, y, _, params = bench_inputs.synthetic_problem(N, d, 4, seed=5 * N + d)
    rng = np.random.default_rng(N)
    ne = d + (1 if kind == 2 else 0)
    ells = np.tile(np.concatenate([np.broadcast_to(params["k_length"], (d,)), [2.3]])[:ne], (B, 1)) * rng.uniform(0.8, 1.25, (B, ne))
    scales = params["k_scale"] * rng.uniform(0.8, 1.25, B)
    noises = params["noise"] * rng.uniform(0.5, 2.0, B)
    noises[3] = -1.0  # not PD: only this entry is NaN
    yres = y[None, :] + 0.05 * rng.standard_normal((B, N))
    engine.set_train(X)
    for yr in (y, yres):
        lml, info, grad, alpha = engine.fit_batch(kind, ells, scales, noises, 1e-6, yr)
        assert info[3] != 0 and np.isnan(lml[3]) and np.all(np.isnan(grad[3]))
        for b in range(B):
            if b == 3:
                continue
            l1, i1 = engine.factor(kind, ells[b], scales[b], noises[b], 1e-6, yr if yr.ndim == 1 else yr[b])
            g_ell, g_s, g_n, a1 = engine.lml_grad()
            assert i1 == 0 and info[b] == 0
            assert l1 == lml[b]
            np.testing.assert_array_equal(np.concatenate([g_ell, [g_s, g_n]]), grad[b])
            np.testing.assert_array_equal(a1, alpha[b])
        lml2, info2, g2, a2 = engine.fit_batch(kind, ells, scales, noises, 1e-6, yr, want_grad=False)
        assert g2 is None and a2 is None
        np.testing.assert_array_equal(lml2[[0, 1, 2, 4]], lml[[0, 1, 2, 4]])


@pytest.mark.parametrize("kind,name", KINDS)
@pytest.mark.parametrize("M,ms", [(300, 100), (300, 128), (257, 100), (90, 200)])
def test_sliced_sweep_equals_slice_by_slice_sweeps(engine, kind, name, M, ms):
    # predict_in_batches semantics inside one sweep: covariance blocks of `ms` test points share ONE factorisation
    # per sample; every block must equal the sweep run on that slice of X_new alone
    N, d, S, n = 260, 2, 7, 2
    X, y, Xn, params = bench_inputs.synthetic_problem(N, d, M, seed=M + ms)
    th = bench_inputs.synthetic_theta_samples(S, d, seed=3)
    rng = np.random.default_rng(4)
    eps = rng.standard_normal((S, n, M))
    yres = y[None, :] + 0.01 * rng.standard_normal((S, N))
    engine.set_train(X)
    means, draws, infos, vars_ = engine.predict_sweep(kind, th["k_length"], th["k_scale"], th["noise"], yres, Xn, False,
                                                      1e-6, eps, want_var=True, m_slice=ms)
    assert np.all(infos == 0) and means.shape == (S, M) and draws.shape == (S, n, M)
    for m0 in range(0, M, ms):
        sl = slice(m0, min(m0 + ms, M))
        m1, d1, i1, v1 = engine.predict_sweep(kind, th["k_length"], th["k_scale"], th["noise"], yres, Xn[sl], False, 1e-6,
                                              np.ascontiguousarray(eps[:, :, sl]), want_var=True)
        np.testing.assert_array_equal(means[:, sl], m1)
        np.testing.assert_array_equal(vars_[:, sl], v1)
        # the block's covariance tiles start at a different row of the ride-along matrix: same arithmetic per element
        np.testing.assert_array_equal(draws[:, :, sl], d1)
    s = 2
    q = {"k_length": th["k_length"][s], "k_scale": th["k_scale"][s], "noise": th["noise"][s]}
    sl = slice(0, min(ms, M))
    m_ref, c_ref = ref.get_mvn_posterior(X, yres[s], Xn[sl], q, False, kernel=name, jitter=1e-6, route="inv")
    assert relerr(means[s, sl], m_ref) < 1e-8
    assert relerr(draws[s][:, sl], ref.mvn_sample(m_ref, c_ref, eps[s][:, sl])) < 1e-6


def test_serialised_trailing_updates_give_the_same_bits(engine):
    """gpx_debug_set_serialise_trailing (bench.py's live `serialised_frac`): the trailing updates of the blocked Cholesky
    (gp.py:160-164) run alone on the chip instead of beside the panel chain — a change of overlap only, so the
    factorisation, its lml and the posterior are the same bits, and the same launches are counted."""
    from gpax_amd import _lib
    N, d = 6400, 2  # 50 tile rows: blocked (two streams, trailing class launches), not a multiple of the outer block
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, 64, seed=3)
    engine.set_train(X)

    def run():
        engine.profile_enable(True)
        engine.profile_reset()
        lml, info = engine.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        n, ms, flops = engine.profile_read(_lib.PROF_GEMM_TRAILING)
        engine.profile_enable(False)
        mean, cov, _ = engine.posterior(Xn, p["noise"], 1e-6, want_cov=True, want_var=False)
        return lml, info, n, flops, mean, cov

    a = run()
    engine.set_serialise_trailing(True)
    try:
        b = run()
    finally:
        engine.set_serialise_trailing(False)
    c = run()
    assert a[1] == 0 and a[2] > 0
    for other in (b, c):
        assert other[0] == a[0] and other[1] == 0 and other[2] == a[2] and other[3] == a[3]
        np.testing.assert_array_equal(other[4], a[4])
        np.testing.assert_array_equal(other[5], a[5])
