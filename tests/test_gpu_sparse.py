"""GPU parity of the variational sparse GP (viSparseGP rows A19-A21) vs the oracle and the golden fixtures."""
import os

import numpy as np
import pytest

from oracle import cpu_ref as ref

import bench_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
KINDS = [(0, "RBF"), (1, "Matern")]


@pytest.mark.parametrize("kind,name", KINDS)
def test_sparse_bound_and_posterior_golden(engine, kind, name):
    g = np.load(os.path.join(GOLD, "sparse.npz"))
    th = g["theta"]
    engine.set_train(g["X"])
    bound, info, _ = engine.sgp_bound(kind, th[:-2], th[-2], th[-1], 1e-6, g["Xu"], g["y"], want_grad=False)
    assert info == 0
    expect = float(g[f"{name}_bound"])
    assert abs(bound - expect) <= 1e-9 * abs(expect)
    for noiseless in [0, 1]:
        noise_p = 0.0 if noiseless else th[-1]
        mean, cov, var, info = engine.sgp_posterior(kind, th[:-2], th[-2], th[-1], 1e-6, g["Xu"], g["y"], g["Xn"],
                                                    noise_p, want_cov=True, want_var=True)
        assert info == 0
        m_ref, c_ref = g[f"{name}_mean_{noiseless}"], g[f"{name}_cov_{noiseless}"]
        # Kuu carries only the 1e-6 jitter (cond ~1e7): agreement is bounded by that conditioning
        assert np.linalg.norm(mean - m_ref) <= 1e-7 * np.linalg.norm(m_ref)
        assert np.abs(cov - c_ref).max() <= 1e-7 * th[-2]
        assert np.abs(var - np.diag(c_ref)).max() <= 1e-7 * th[-2]


@pytest.mark.parametrize("kind,name", KINDS)
@pytest.mark.parametrize("N,d,Mi", [(150, 1, 12), (400, 2, 40), (700, 3, 150)])
def test_sparse_gradient_matches_finite_differences_of_oracle(engine, kind, name, N, d, Mi):
    X, y, _, p = bench_inputs.synthetic_problem(N, d, 4, seed=N + Mi)
    rng = np.random.default_rng(1)
    Xu = X[rng.choice(N, Mi, replace=False)] + 0.05 * rng.standard_normal((Mi, d))
    engine.set_train(X)
    jit = 1e-4  # keeps cond(Kuu) moderate so that central differences of the oracle are meaningful
    bound, info, g = engine.sgp_bound(kind, p["k_length"], p["k_scale"], p["noise"], jit, Xu, y)
    assert info == 0
    f0 = ref.sparse_bound(X, y, Xu, p, kernel=name, jitter=jit)
    assert abs(bound - f0) <= 1e-8 * abs(f0)

    def f(ell, scale, noise, xu, yy=y):
        return ref.sparse_bound(X, yy, xu, {"k_length": ell, "k_scale": scale, "noise": noise}, kernel=name, jitter=jit)

    ell0, s0, n0 = np.asarray(p["k_length"], dtype=float), p["k_scale"], p["noise"]
    scale_g = max(np.abs(g["k_length"]).max(), abs(g["k_scale"]), abs(g["noise"]))
    for m in range(d):
        h = 1e-5 * ell0[m]
        a, b = ell0.copy(), ell0.copy()
        a[m] += h
        b[m] -= h
        fd = (f(a, s0, n0, Xu) - f(b, s0, n0, Xu)) / (2 * h)
        assert abs(fd - g["k_length"][m]) <= 2e-5 * scale_g
    h = 1e-5 * s0
    assert abs((f(ell0, s0 + h, n0, Xu) - f(ell0, s0 - h, n0, Xu)) / (2 * h) - g["k_scale"]) <= 2e-5 * scale_g
    h = 1e-5 * n0
    assert abs((f(ell0, s0, n0 + h, Xu) - f(ell0, s0, n0 - h, Xu)) / (2 * h) - g["noise"]) <= 2e-5 * scale_g
    gx_scale = np.abs(g["Xu"]).max()
    for (a_, m_) in [(0, 0), (Mi // 2, d - 1), (Mi - 1, 0)]:
        h = 1e-5
        xp, xm = Xu.copy(), Xu.copy()
        xp[a_, m_] += h
        xm[a_, m_] -= h
        fd = (f(ell0, s0, n0, xp) - f(ell0, s0, n0, xm)) / (2 * h)
        assert abs(fd - g["Xu"][a_, m_]) <= 5e-5 * max(gx_scale, 1.0)
    for n_ in [0, N // 3]:
        h = 1e-5
        yp, ym = y.copy(), y.copy()
        yp[n_] += h
        ym[n_] -= h
        fd = (f(ell0, s0, n0, Xu, yp) - f(ell0, s0, n0, Xu, ym)) / (2 * h)
        assert abs(fd - g["yres"][n_]) <= 1e-5 * max(np.abs(g["yres"]).max(), 1.0)


def test_sparse_c5_shape_runs_and_is_deterministic(engine):
    # a scaled-down C5: image-like 2-D inputs, Matern, inducing ratio 0.125
    rng = np.random.default_rng(3)
    N, Mi = 4096, 512
    X = rng.uniform(0, 64, (N, 2))
    y = np.sin(X[:, 0] / 7) * np.cos(X[:, 1] / 5) + 0.05 * rng.standard_normal(N)
    Xu = X[rng.choice(N, Mi, replace=False)]
    engine.set_train(X)
    b1, info, g1 = engine.sgp_bound(1, [8.0, 8.0], 1.0, 0.01, 1e-6, Xu, y)
    b2, _, g2 = engine.sgp_bound(1, [8.0, 8.0], 1.0, 0.01, 1e-6, Xu, y)
    assert info == 0 and np.isfinite(b1) and b1 == b2
    np.testing.assert_array_equal(g1["Xu"], g2["Xu"])
    full = ref.exactgp_log_likelihood(X, y, {"k_length": np.array([8.0, 8.0]), "k_scale": 1.0, "noise": 0.01},
                                      kernel="Matern")
    assert b1 <= full + 1e-6 * abs(full)  # the VFE bound never exceeds the exact log marginal likelihood


@pytest.mark.parametrize("N,Mi", [(900, 100), (1500, 300), (2100, 700)])
def test_solves_by_inverse_agree_with_the_sweeps(monkeypatch, N, Mi):
    """sparse.hip round 3: W = Kfu Luu^-T, V1 = Ksu Luu^-T, V2 = V1 LA^-T as ONE GEMM each against L^-1 (from the
    L^-T tree) and the gradient products trimmed to the triangle of Luu^-T / re-associated — against GPX_SGP_SOLVE=sweep
    (the right-looking solves of rounds 1 - 2): bound, every gradient component and the posterior agree to rounding, and
    both agree with the oracle (sparse_gp.py:62-114, 173-223).  Round 5: so does GPX_SGP_SOLVE=ride, where W is the blocked
    right-looking solve that follows the Cholesky chain of Kuu group by group on three streams (M = 700: groups of 2, 2 and
    a ragged last one; M = 100: a single group)."""
    from gpax_amd import _lib
    X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, 150, seed=5)
    Xu = X[np.random.default_rng(2).choice(N, Mi, replace=False)] + 1e-3
    outs = {}
    for mode in ("inverse", "sweep", "ride"):
        monkeypatch.setenv("GPX_SGP_SOLVE", mode)
        e = _lib.Engine(0)
        e.set_train(X)
        b, info, g = e.sgp_bound(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, Xu, y, True)
        mean, cov, var, info2 = e.sgp_posterior(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, Xu, y, Xn, 0.0,
                                                want_cov=True, want_var=True)
        assert info == 0 and info2 == 0
        outs[mode] = (b, g, mean, cov, var)
        e.close()
    s = outs["sweep"]
    for a in (outs["inverse"], outs["ride"]):
        assert abs(a[0] - s[0]) <= 1e-10 * abs(s[0])
        for k in ("k_length", "k_scale", "noise", "Xu", "yres"):
            u, v = np.asarray(a[1][k], dtype=float), np.asarray(s[1][k], dtype=float)
            np.testing.assert_allclose(u, v, rtol=1e-7, atol=1e-8 * max(1.0, np.abs(v).max()))
        np.testing.assert_allclose(a[2], s[2], rtol=0, atol=1e-9 * np.abs(s[2]).max())
        np.testing.assert_allclose(a[3], s[3], rtol=0, atol=1e-9 * np.abs(s[3]).max())
    a = outs["inverse"]
    expect = ref.sparse_bound(X, y, Xu, p, kernel="Matern", jitter=1e-6)
    assert abs(a[0] - expect) <= 1e-8 * abs(expect)
    m_ref, c_ref = ref.sparse_posterior(X, y, Xu, Xn, p, noiseless=True, kernel="Matern", jitter=1e-6)
    np.testing.assert_allclose(a[2], m_ref, rtol=0, atol=1e-7 * np.abs(m_ref).max())
    np.testing.assert_allclose(a[3], c_ref, rtol=0, atol=1e-7 * np.abs(c_ref).max())


@pytest.mark.parametrize("kind,name", KINDS)
def test_clipped_trace_term_value_and_gradient(engine, kind, name):
    """sparse_gp.py:100-104: trace_term = clip(N kd - |W|_F^2, a_min=0) — zero value AND zero gradient when clipped.  The
    switch is evaluated on the device (mat_combine_kernel) and again on the host for the scale / noise terms (no
    test exercised it).  Forced here with inducing points AT well separated training points and a NEGATIVE jitter:
    Kuu = Kff - 1e-3 I (still positive definite) makes Qff = Kff Kuu^-1 Kff exceed Kff, so N kd - |W|_F^2 < 0.  Bound
    against the oracle, gradients against central differences of the oracle (whose clip has the same zero slope)."""
    rng = np.random.default_rng(4)
    g1 = np.arange(12, dtype=float)
    X = np.stack(np.meshgrid(g1, g1, indexing="ij"), axis=-1).reshape(-1, 2) + 0.05 * rng.standard_normal((144, 2))
    X[-1] = X[-2] + 0.01  # the one training point without an inducing point of its own sits next to one
    y = np.sin(X[:, 0]) + 0.1 * rng.standard_normal(144)
    Xu = X[:-1].copy()  # (all but one: with M = N the same-shape rule also puts 1e-6 on the diagonal of Kuf — next test)
    ell, scale, noise, jit = np.array([0.35, 0.4]), 1.2, 0.05, -1e-3
    p = {"k_length": ell, "k_scale": scale, "noise": noise}
    engine.set_train(X)
    bound, info, g = engine.sgp_bound(kind, ell, scale, noise, jit, Xu, y)
    assert info == 0

    def f(ell_, scale_, noise_, xu=Xu):
        return ref.sparse_bound(X, y, xu, {"k_length": ell_, "k_scale": scale_, "noise": noise_}, kernel=name, jitter=jit)

    f0 = f(ell, scale, noise)
    assert abs(bound - f0) <= 1e-9 * abs(f0)
    # the clip IS active here: the raw trace term of the reference's formula (sparse_gp.py:96-104) is negative
    kfn = ref.get_kernel(name)
    import scipy.linalg as sla
    W = sla.solve_triangular(np.linalg.cholesky(kfn(Xu, Xu, p, jitter=jit)), kfn(Xu, X, p), lower=True).T
    assert (np.diag(kfn(X, X, p, jitter=0)) - np.square(W).sum(axis=-1)).sum() < -1e-3
    scale_g = max(np.abs(g["k_length"]).max(), abs(g["k_scale"]), abs(g["noise"]))
    for m in range(2):
        h = 1e-6 * ell[m]
        a, b = ell.copy(), ell.copy()
        a[m] += h
        b[m] -= h
        assert abs((f(a, scale, noise) - f(b, scale, noise)) / (2 * h) - g["k_length"][m]) <= 5e-5 * scale_g
    h = 1e-6 * scale
    assert abs((f(ell, scale + h, noise) - f(ell, scale - h, noise)) / (2 * h) - g["k_scale"]) <= 5e-5 * scale_g
    h = 1e-6 * noise
    assert abs((f(ell, scale, noise + h) - f(ell, scale, noise - h)) / (2 * h) - g["noise"]) <= 5e-5 * scale_g
    gx = np.abs(g["Xu"]).max()
    for a_, m_ in [(0, 0), (77, 1)]:
        h = 1e-6
        xp, xm = Xu.copy(), Xu.copy()
        xp[a_, m_] += h
        xm[a_, m_] -= h
        assert abs((f(ell, scale, noise, xp) - f(ell, scale, noise, xm)) / (2 * h) - g["Xu"][a_, m_]) <= 1e-4 * max(gx, 1.0)


def test_as_many_inducing_points_as_training_points_follow_the_same_shape_rule(engine):
    """kernels.py:63-65 adds (noise + jitter) I iff the two inputs have the same SHAPE, and viSparseGP.model builds Kuf with
    the kernel's default jitter (sparse_gp.py:96): at inducing ratio 1 the bound's Kuf carries 1e-6 on its diagonal,
    get_mvn_posterior's (jitter=0, sparse_gp.py:196) does not.  Both against the oracle, which restates the calls as they
    are; and the cached forward pass of the one is not reused for the other."""
    rng = np.random.default_rng(8)
    N = 96
    X = rng.uniform(0, 10, (N, 2))
    y = np.sin(X[:, 0]) * np.cos(X[:, 1]) + 0.05 * rng.standard_normal(N)
    Xu = X[rng.permutation(N)] + 0.2 * rng.standard_normal((N, 2))
    Xn = rng.uniform(0, 10, (40, 2))
    p = {"k_length": np.array([1.5, 2.0]), "k_scale": 1.1, "noise": 0.02}
    jit = 1e-2  # large against the 1e-6 under test, so that the comparison resolves it
    engine.set_train(X)
    for _ in range(2):  # bound -> posterior -> bound -> posterior: never the other call's forward pass
        bound, info, _ = engine.sgp_bound(1, p["k_length"], p["k_scale"], p["noise"], jit, Xu, y, want_grad=False)
        expect = ref.sparse_bound(X, y, Xu, p, kernel="Matern", jitter=jit)
        assert info == 0 and abs(bound - expect) <= 1e-11 * abs(expect)
        mean, cov, _, info = engine.sgp_posterior(1, p["k_length"], p["k_scale"], p["noise"], jit, Xu, y, Xn, 0.0,
                                                  want_cov=True, want_var=False)
        m_ref, c_ref = ref.sparse_posterior(X, y, Xu, Xn, p, noiseless=True, kernel="Matern", jitter=jit)
        assert info == 0 and np.abs(mean - m_ref).max() <= 1e-10 * np.abs(m_ref).max()
        assert np.abs(cov - c_ref).max() <= 1e-10 * np.abs(c_ref).max()
    # the rule makes a difference the test can see: the bound WITHOUT the diagonal term differs by more than the tolerance
    kfn = ref.get_kernel("Matern")
    import scipy.linalg as sla
    Luu = np.linalg.cholesky(kfn(Xu, Xu, p, jitter=jit))
    W0 = sla.solve_triangular(Luu, kfn(Xu, X, p, jitter=0), lower=True).T
    W1 = sla.solve_triangular(Luu, kfn(Xu, X, p), lower=True).T
    assert np.abs(W0 - W1).max() > 1e-8


def test_forward_pass_is_reused_only_for_identical_inputs():
    """sparse.hip sgp_setup: the forward pass (Kuu, Kfu, both factorisations, c) of the last call is kept while the next
    call brings bit-identical kernel / theta / jitter / Xu / yres on the same X — what predict_in_batches does slice after
    slice (sparse_gp.py:173-223 redoes it per slice).  Reused or not, the values are those of a fresh context; a new
    yres, theta or X is noticed."""
    from gpax_amd import _lib
    X, y, Xn, p = bench_inputs.synthetic_problem(1100, 2, 120, seed=9)
    Xu = X[::7] + 1e-3
    args = lambda yy, ell: (1, ell, p["k_scale"], p["noise"], 1e-6, Xu, yy)  # noqa: E731

    def fresh(Xt, yy, ell, Xs):
        e = _lib.Engine(0)
        e.set_train(Xt)
        out = e.sgp_posterior(*args(yy, ell), Xs, 0.0, want_cov=False, want_var=True)
        e.close()
        return out

    e = _lib.Engine(0)
    e.set_train(X)
    e.sgp_posterior(*args(y, p["k_length"]), Xn[:50], 0.0, want_cov=False, want_var=True)
    again = e.sgp_posterior(*args(y, p["k_length"]), Xn, 0.0, want_cov=False, want_var=True)  # forward pass reused
    base = fresh(X, y, p["k_length"], Xn)
    np.testing.assert_array_equal(again[0], base[0])
    np.testing.assert_array_equal(again[2], base[2])
    # (bound without gradient -> posterior -> bound WITH gradient on the one cached forward pass — the |W|_F^2
    # partials the gradient's clipped-trace switch reads were left by the forward pass, not by this call)
    b0, _, _ = e.sgp_bound(*args(y, p["k_length"]), False)
    e.sgp_posterior(*args(y, p["k_length"]), Xn[:30], 0.0, want_cov=False, want_var=True)
    b1, _, g1 = e.sgp_bound(*args(y, p["k_length"]), True)  # reused by the bound and its gradient as well
    assert b0 == b1
    e2 = _lib.Engine(0)
    e2.set_train(X)
    b2, _, g2 = e2.sgp_bound(*args(y, p["k_length"]), True)
    e2.close()
    assert b1 == b2
    for k in g1:
        np.testing.assert_array_equal(np.asarray(g1[k]), np.asarray(g2[k]))
    y2 = y + 0.25
    np.testing.assert_array_equal(e.sgp_posterior(*args(y2, p["k_length"]), Xn, 0.0, want_cov=False)[0], fresh(X, y2, p["k_length"], Xn)[0])
    ell2 = 1.3 * np.asarray(p["k_length"])
    np.testing.assert_array_equal(e.sgp_posterior(*args(y2, ell2), Xn, 0.0, want_cov=False)[0], fresh(X, y2, ell2, Xn)[0])
    X2 = X + 0.01
    e.set_train(X2)
    np.testing.assert_array_equal(e.sgp_posterior(*args(y2, ell2), Xn, 0.0, want_cov=False)[0], fresh(X2, y2, ell2, Xn)[0])
    e.close()


def test_vector_valued_mean_site_on_the_gpu():
    """viSparseGP with a 2-element mean-function site (sparse_gp.py:85-89): the gradient of the objective w.r.t. each
    element against central differences of the device objective, the posterior against the oracle (mean function
    subtracted from the residual and added back at X_new, sparse_gp.py:189-192,219-221)."""
    from bench_inputs import synthetic_problem
    from gpax_amd import _lib, dist, plate, sample, viSparseGP
    from oracle import cpu_ref as ref

    _lib.set_engine(None)
    X, y, Xn, _ = synthetic_problem(400, 2, 50, seed=11)
    y = y + 0.5 * X[:, 0] - 0.3 * X[:, 1]

    def mean_fn(x, p):
        return p["w"][0] * x[:, 0] + p["w"][1] * x[:, 1]

    def mean_prior():
        with plate("w_plate", 2):
            w = sample("w", dist.Normal(0.0, 2.0))
        return {"w": w}

    m = viSparseGP(2, "Matern", mean_fn=mean_fn, mean_fn_prior=mean_prior)
    m.X_train, m.y_train = m._set_data(X, y)
    sites = m._sites()
    nu = sum(s.size for s in sites)
    rng = np.random.default_rng(1)
    Xu = X[rng.choice(400, 40, replace=False)].copy()
    x0 = np.concatenate([0.2 * rng.standard_normal(nu), Xu.reshape(-1)])
    val, grad = m._sparse_log_joint(sites, x0, 40, 1e-4, jacobian=False)
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
        fd = (m._sparse_log_joint(sites, xp, 40, 1e-4, False)[0] - m._sparse_log_joint(sites, xm, 40, 1e-4, False)[0]) / (2 * h)
        assert abs(grad[i] - fd) <= 2e-6 * max(1.0, abs(fd)), (i, grad[i], fd)
    m.Xu = Xu
    params = {"k_length": np.array([1.3, 1.6]), "k_scale": 1.1, "noise": 0.05, "w": np.array([0.45, -0.25])}
    mean, cov = m.get_mvn_posterior(Xn, params, jitter=1e-4)
    e_mean, e_cov = ref.sparse_posterior(X, y, Xu, Xn, params, False, kernel="Matern", jitter=1e-4, mean_fn=mean_fn,
                                         mean_fn_has_params=True)
    assert np.linalg.norm(mean - e_mean) <= 1e-8 * np.linalg.norm(e_mean)
    assert np.linalg.norm(cov - e_cov) <= 1e-8 * np.linalg.norm(e_cov)
