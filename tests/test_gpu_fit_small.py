"""The fused small-N fit step (csrc/fit_small.hip: Gram + augmentation + factorisation + lml terms + alpha + K^-1 + gradient
contraction as ONE launch, N <= 128) against the general launch sequence (GPX_FIT_SMALL=0) and the oracle
(gpax/models/gp.py:137-164 and its reverse-mode gradient): every N where a branch of the kernel changes — one tile row, the
LDS-resident factorisation up to N = 63, the 128 x 128 kernel above — all three kernels, generic d, failed pivots."""
import numpy as np
import pytest

import bench_inputs
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu

KINDS = [(0, "RBF"), (1, "Matern"), (2, "Periodic")]


def _problem(N, d, kind, seed):
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, 11, seed=seed)
    ell = np.asarray(p["k_length"], dtype=float)
    if kind == 2:
        ell = np.concatenate([ell, [2.7]])  # period rides behind the length scales (include/gpx.h)
    return X, y, Xn, ell, p["k_scale"], p["noise"]


def _run(monkeypatch, fused, X, y, Xn, kind, ell, scale, noise):
    from gpax_amd import _lib
    monkeypatch.setenv("GPX_FIT_SMALL", "1" if fused else "0")
    e = _lib.Engine(0)
    e.set_train(X)
    lml, info = e.factor(kind, ell, scale, noise, 1e-6, y)
    g_ell, g_s, g_n, alpha = e.lml_grad()
    lml2, _ = e.factor(kind, ell, scale, noise, 1e-6, y)
    mean, cov, var = e.posterior(Xn, noise, 1e-6, want_cov=True, want_var=True)
    ells = np.stack([ell, 1.07 * ell, 0.93 * ell])
    fb = e.fit_batch(kind, ells, [scale, 1.1 * scale, scale], [noise, noise, 2 * noise], 1e-6, y)
    fb1 = e.fit_batch(kind, ells[:1], [scale], [noise], 1e-6, y)
    e.close()
    return dict(lml=lml, info=info, grad=np.concatenate([g_ell, [g_s, g_n]]), alpha=alpha, lml2=lml2, mean=mean, cov=cov,
                var=var, fb=fb, fb1=fb1)


@pytest.mark.parametrize("kind,name", KINDS)
@pytest.mark.parametrize("N,d", [(2, 1), (7, 1), (15, 2), (16, 1), (25, 1), (31, 3), (32, 2), (47, 1), (63, 2), (64, 1), (100, 5),
                                 (127, 2), (128, 1), (128, 3)])
def test_fused_fit_step_equals_the_general_path_and_the_oracle(monkeypatch, kind, name, N, d):
    X, y, Xn, ell, scale, noise = _problem(N, d, kind, seed=100 * N + d)
    f = _run(monkeypatch, True, X, y, Xn, kind, ell, scale, noise)
    g = _run(monkeypatch, False, X, y, Xn, kind, ell, scale, noise)
    assert f["info"] == 0 and g["info"] == 0
    # the factor and its inverse are bit for bit the general path's (same Gram arithmetic, same tile operations): what is read
    # off them afterwards — the posterior — is identical; the reductions of the lml and the gradient run in another order
    np.testing.assert_array_equal(f["mean"], g["mean"])
    np.testing.assert_array_equal(f["cov"], g["cov"])
    np.testing.assert_array_equal(f["var"], g["var"])
    assert abs(f["lml"] - g["lml"]) <= 1e-13 * max(1.0, abs(g["lml"])) and f["lml"] == f["lml2"]
    np.testing.assert_allclose(f["alpha"], g["alpha"], rtol=0, atol=1e-13 * max(1.0, np.abs(g["alpha"]).max()))
    np.testing.assert_allclose(f["grad"], g["grad"], rtol=0, atol=1e-11 * max(1.0, np.abs(g["grad"]).max()))
    # the batched call (hyper-parameters, residuals and results through page-locked memory) and the single-theta pair
    # give the same bits, entry by entry
    assert f["fb"][0][0] == f["lml"] and f["fb1"][0][0] == f["lml"]
    np.testing.assert_array_equal(f["fb"][2][0], f["grad"])
    np.testing.assert_array_equal(f["fb1"][2][0], f["grad"])
    np.testing.assert_array_equal(f["fb"][3][0], f["alpha"])
    np.testing.assert_allclose(f["fb"][0], g["fb"][0], rtol=1e-13)
    np.testing.assert_allclose(f["fb"][2], g["fb"][2], rtol=0, atol=1e-11 * max(1.0, np.abs(g["fb"][2]).max()))
    if kind != 2:  # the oracle's analytic gradient covers RBF / Matern (Periodic: test_gpu_periodic.py's finite differences)
        p = {"k_length": ell, "k_scale": scale, "noise": noise}
        expect = ref.exactgp_log_likelihood(X, y, p, kernel=name)
        assert abs(f["lml"] - expect) <= 1e-10 * max(1.0, abs(expect))
        e_ell, e_scale, e_noise, _ = ref.exactgp_log_likelihood_grad(X, y, p, kernel=name)
        want = np.concatenate([np.atleast_1d(e_ell), [e_scale, e_noise]])
        np.testing.assert_allclose(f["grad"], want, rtol=0, atol=1e-8 * max(1.0, np.abs(want).max()))


def test_fused_fit_step_reports_a_failed_pivot_like_the_general_path(monkeypatch):
    """tests/test_gp.py:196-206 of the reference feeds hyper-parameters that make K indefinite: NaN and a pivot report, never a
    crash — from the LDS-resident factorisation (N = 30) and from the 128 x 128 kernel (N = 90, and N = 128 where the augmentation row rides below the block) alike."""
    from gpax_amd import _lib
    for N in (30, 90, 128):
        X, y, _, p = bench_inputs.synthetic_problem(N, 1, 4, seed=N)
        X[N // 2] = X[N // 2 - 1]  # two coincident points and a negative "noise": a non-positive pivot
        out = {}
        for fused in ("1", "0"):
            monkeypatch.setenv("GPX_FIT_SMALL", fused)
            e = _lib.Engine(0)
            e.set_train(X)
            lml, info = e.factor(0, p["k_length"], p["k_scale"], -0.5, 1e-6, y)
            fb = e.fit_batch(0, np.stack([p["k_length"]] * 2), [p["k_scale"]] * 2, [-0.5, 0.1], 1e-6, y)
            out[fused] = (lml, info, fb)
            e.close()
        assert np.isnan(out["1"][0]) and out["1"][1] == out["0"][1] > 0
        np.testing.assert_array_equal(out["1"][2][1], out["0"][2][1])  # pivot reports of the batch: [failed, fine]
        assert np.isnan(out["1"][2][0][0]) and np.isfinite(out["1"][2][0][1]) and np.all(np.isnan(out["1"][2][2][0]))


def test_fused_fit_step_with_a_per_point_diagonal_takes_the_general_path(monkeypatch):
    """MeasuredNoiseGP / VarNoiseGP set a per-point variance (gpx_set_diag) and read d lml / d v from K^-1, which the fused
    kernel never stores: those contexts keep the general sequence, whatever N."""
    from gpax_amd import _lib
    X, y, _, p = bench_inputs.synthetic_problem(40, 2, 4, seed=3)
    v = 0.05 + 0.1 * np.random.default_rng(0).uniform(size=40)
    e = _lib.Engine(0)
    e.set_train(X)
    e.set_diag(v)
    lml, info = e.factor(1, p["k_length"], p["k_scale"], 0.0, 1e-6, y)
    e.lml_grad()
    gd = e.lml_grad_diag()
    e.close()
    Kmat = ref.get_kernel("Matern")(X, X, p, 0.0, 1e-6) + np.diag(v)
    Ki = np.linalg.inv(Kmat)
    al = Ki @ y
    np.testing.assert_allclose(gd, 0.5 * (al * al - np.diag(Ki)), rtol=0, atol=1e-9 * np.abs(al).max() ** 2)
    assert info == 0 and np.isfinite(lml)


def test_lml_grad_diag_after_the_one_launch_step_reruns_the_general_sequence():
    """without gpx_set_diag the fit step of N <= 128 is ONE launch that never stores K^-1; gpx_lml_grad_diag
    (d lml / d v, hskgp.py:124-153) called behind it used to fail with 'must follow gpx_lml_grad'.  It now re-runs the general
    sequence at the same theta: 1/2 (alpha_i^2 - (K^-1)_ii) against the oracle's dense inverse."""
    from gpax_amd import _lib
    for N in (40, 128):
        X, y, _, p = bench_inputs.synthetic_problem(N, 2, 4, seed=N)
        e = _lib.Engine(0)
        e.set_train(X)
        lml, info = e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        g = e.lml_grad()
        gd = e.lml_grad_diag()
        g2 = (e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y), e.lml_grad())[1]  # the fused path still in force
        e.close()
        Ki = np.linalg.inv(ref.get_kernel("Matern")(X, X, p, p["noise"], 1e-6))
        al = Ki @ y
        np.testing.assert_allclose(gd, 0.5 * (al * al - np.diag(Ki)), rtol=0, atol=1e-9 * max(1.0, np.abs(al).max() ** 2))
        for a, b in zip(g, g2):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
