"""Seeded randomised parity sweep: random sizes, input dimensions, kernels, batch sizes, covariance-block sizes and
residual layouts through gpx_predict_sweep / gpx_fit_batch, each case against the oracle."""
import numpy as np
import pytest

from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu

NAMES = {0: "RBF", 1: "Matern", 2: "Periodic"}


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    kind = int(rng.integers(0, 3))
    N = int(rng.choice([2, 3, 17, 64, 127, 128, 129, 200, 333, 511, 640]))
    M = int(rng.choice([1, 2, 31, 100, 128, 129, 257]))
    d = int(rng.integers(1, 7))
    S = int(rng.integers(1, 9))
    n = int(rng.integers(0, 4))
    X = rng.uniform(0, 5, (N, d))
    Xn = rng.uniform(0, 5, (M, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(N)
    ne = d + (1 if kind == 2 else 0)
    ells = rng.uniform(0.7, 2.0, (S, ne))
    if kind == 2:
        ells[:, d] = rng.uniform(2.0, 4.0, S)
    scales, noises = rng.uniform(0.5, 2.0, S), rng.uniform(0.05, 0.5, S)
    strided = bool(rng.integers(0, 2))
    yres = y[None, :] + 0.02 * rng.standard_normal((S, N)) if strided else y
    eps = rng.standard_normal((S, n, M)) if n else None
    batch = int(rng.choice([0, 1, 2, 3]))
    m_slice = int(rng.choice([0, 0, 1, 50, 128, 130]))
    noiseless = bool(rng.integers(0, 2))
    return kind, X, Xn, yres, ells, scales, noises, eps, n, batch, m_slice, noiseless, d


@pytest.mark.parametrize("seed", range(40))
def test_random_sweep_case_matches_oracle(engine, seed, monkeypatch):
    kind, X, Xn, yres, ells, scales, noises, eps, n, batch, m_slice, noiseless, d = _case(seed)
    name = NAMES[kind]
    S, M = ells.shape[0], Xn.shape[0]
    monkeypatch.setenv("GPX_SWEEP_BATCH", str(batch))
    engine.set_train(X)
    means, draws, infos, vars_ = engine.predict_sweep(kind, ells, scales, noises, yres, Xn, noiseless, 1e-6, eps,
                                                      want_var=True, m_slice=m_slice)
    assert np.all(infos == 0), (seed, infos)
    Ms = m_slice if 0 < m_slice < M else M
    for s in range(S):
        p = {"k_length": ells[s, :d], "k_scale": scales[s], "noise": noises[s]}
        if kind == 2:
            p["period"] = ells[s, d]
        ys = yres if yres.ndim == 1 else yres[s]
        m_ref, c_ref = ref.get_mvn_posterior(X, ys, Xn, p, noiseless, kernel=name, jitter=1e-6, route="inv")
        scale = max(np.linalg.norm(m_ref), 1e-9)
        assert np.linalg.norm(means[s] - m_ref) <= 1e-8 * scale, (seed, s)
        np.testing.assert_allclose(vars_[s], np.diag(c_ref), rtol=1e-7, atol=1e-10)
        if n:
            for m0 in range(0, M, Ms):  # draws are independent per covariance block
                sl = slice(m0, min(m0 + Ms, M))
                d_ref = ref.mvn_sample(m_ref[sl], c_ref[sl, sl], eps[s][:, sl])
                assert np.linalg.norm(draws[s][:, sl] - d_ref) <= 1e-6 * max(np.linalg.norm(d_ref), 1e-9), (seed, s, m0)


@pytest.mark.parametrize("seed", range(12))
def test_random_fit_batch_case_matches_oracle(engine, seed):
    kind, X, _, yres, ells, scales, noises, _, _, _, _, _, d = _case(100 + seed)
    name = NAMES[kind]
    B = ells.shape[0]
    engine.set_train(X)
    lml, info, grad, alpha = engine.fit_batch(kind, ells, scales, noises, 1e-6, yres)
    assert np.all(info == 0)
    for b in range(B):
        p = {"k_length": ells[b, :d], "k_scale": scales[b], "noise": noises[b]}
        if kind == 2:
            p["period"] = ells[b, d]
        ys = yres if yres.ndim == 1 else yres[b]
        expect = ref.exactgp_log_likelihood(X, ys, p, kernel=name, jitter=1e-6)
        assert abs(lml[b] - expect) <= 1e-10 * max(1.0, abs(expect))
        K = ref.get_kernel(name)(X, X, p, p["noise"], jitter=1e-6)
        a_ref = np.linalg.solve(K, ys)
        assert np.linalg.norm(alpha[b] - a_ref) <= 1e-8 * np.linalg.norm(a_ref)
        if kind != 2:
            e_ell, e_s, e_n, _ = ref.exactgp_log_likelihood_grad(X, ys, p, kernel=name, jitter=1e-6, yres=ys)
            sc = max(np.abs(e_ell).max(), abs(e_s), abs(e_n))
            np.testing.assert_allclose(grad[b], np.concatenate([e_ell, [e_s, e_n]]), rtol=1e-7, atol=1e-8 * sc)
