"""GPU vs the committed golden fixtures (tests/golden/*.npz, minted by tools/make_golden.py from the
oracle; no oracle import needed here)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def test_gram_golden(engine):
    g = load("gram")
    for c in range(int(g["ncases"])):
        kind, scale, noise, jitter = g[f"c{c}_meta"]
        X, Z = g[f"c{c}_X"], g[f"c{c}_Z"]
        add = X.shape == Z.shape
        K = engine.gram(int(kind), X, Z, g[f"c{c}_ell"], scale, noise + jitter, add)
        np.testing.assert_allclose(K, g[f"c{c}_K"], rtol=1e-10, atol=1e-12, err_msg=f"case {c}")


def test_lml_and_gradient_golden(engine):
    g = load("lml")
    for c in range(int(g["ncases"])):
        kind, jitter = g[f"c{c}_meta"]
        th = g[f"c{c}_theta"]
        engine.set_train(g[f"c{c}_X"])
        lml, info = engine.factor(int(kind), th[:-2], th[-2], th[-1], jitter, g[f"c{c}_y"])
        assert info == 0
        expect = float(g[f"c{c}_lml"])
        assert abs(lml - expect) <= 1e-10 * abs(expect), c  # |dLML|/|LML| <= 1e-10
        g_ell, g_s, g_n, alpha = engine.lml_grad()
        got = np.concatenate([g_ell, [g_s, g_n]])
        want = g[f"c{c}_grad"]
        np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-8 * np.abs(want).max())
        assert np.linalg.norm(alpha - g[f"c{c}_alpha"]) <= 1e-9 * np.linalg.norm(g[f"c{c}_alpha"])


def test_posterior_and_draw_golden(engine):
    g = load("posterior")
    for c in range(int(g["ncases"])):
        kind, noiseless, jitter = g[f"c{c}_meta"]
        th = g[f"c{c}_theta"]
        engine.set_train(g[f"c{c}_X"])
        engine.factor(int(kind), th[:-2], th[-2], th[-1], jitter, g[f"c{c}_y"])
        noise_p = 0.0 if noiseless else th[-1]
        mean, cov, var = engine.posterior(g[f"c{c}_Xn"], noise_p, jitter, want_cov=True, want_var=True)
        m_ref, c_ref = g[f"c{c}_mean"], g[f"c{c}_cov"]
        assert np.linalg.norm(mean - m_ref) <= 1e-8 * np.linalg.norm(m_ref), c
        scale = th[-2] + noise_p + jitter  # |k_pp| scale
        assert np.abs(cov - c_ref).max() <= 1e-8 * scale, c
        assert np.abs(var - np.diag(c_ref)).max() <= 1e-8 * scale, c
        draws, info = engine.mvn_draw(g[f"c{c}_eps"])
        d_ref = g[f"c{c}_draws"]
        if np.isnan(d_ref).any():
            continue  # noiseless + tiny jitter: the reference's cov may be numerically non-PD
        assert info == 0
        # chol of a nearly singular posterior cov amplifies the 1e-13 cov differences: tolerance by cond
        tol = 1e-6 if noiseless else 1e-8
        assert np.linalg.norm(draws - d_ref) <= tol * np.linalg.norm(d_ref), c


@pytest.mark.parametrize("kind,name", [(0, "RBF"), (1, "Matern")])
def test_sweep_golden(engine, kind, name):
    g = load("sweep")
    engine.set_train(g["X"])
    means, draws, infos = engine.predict_sweep(kind, g["s_k_length"], g["s_k_scale"], g["s_noise"], g["y"], g["Xn"],
                                               False, 1e-6, g["eps"])
    assert np.all(infos == 0)
    assert np.linalg.norm(means - g[f"{name}_means"]) <= 1e-8 * np.linalg.norm(g[f"{name}_means"])
    assert np.linalg.norm(draws - g[f"{name}_y_sampled"]) <= 1e-8 * np.linalg.norm(g[f"{name}_y_sampled"])
    np.testing.assert_allclose(means.mean(0), g[f"{name}_mean_of_means"], rtol=1e-8, atol=1e-10)
