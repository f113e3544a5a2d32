"""GPU edge cases of the hot path through the C-ABI: tiny and ragged sizes, the maximum input dimension,
means-only sweeps, near-singular and non-PD hyper-parameters (NaN rows, never a crash)."""
import numpy as np
import pytest

from oracle import cpu_ref as ref

import bench_inputs

pytestmark = pytest.mark.gpu


def _check_posterior(engine, kind, name, X, y, Xn, p, tol=1e-8):
    engine.set_train(X)
    lml, info = engine.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    assert info == 0
    expect = ref.exactgp_log_likelihood(X, y, p, kernel=name, jitter=1e-6)
    assert abs(lml - expect) <= 1e-10 * max(1.0, abs(expect))
    mean, cov, var = engine.posterior(Xn, p["noise"], 1e-6, want_cov=True, want_var=True)
    m_ref, c_ref = ref.get_mvn_posterior(X, y, Xn, p, False, kernel=name, jitter=1e-6, route="inv")
    assert np.linalg.norm(mean - m_ref) <= tol * max(np.linalg.norm(m_ref), 1e-12)
    assert np.linalg.norm(cov - c_ref) <= tol * np.linalg.norm(c_ref)
    np.testing.assert_allclose(var, np.diag(c_ref), rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("kind,name", [(0, "RBF"), (1, "Matern")])
@pytest.mark.parametrize("N,M", [(2, 1), (3, 5), (2, 130), (127, 1), (129, 129)])
def test_tiny_and_ragged_sizes(engine, kind, name, N, M):
    rng = np.random.default_rng(N * 131 + M)
    X, Xn = rng.uniform(0, 4, (N, 2)), rng.uniform(0, 4, (M, 2))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(N)
    p = {"k_length": np.array([1.0, 1.4]), "k_scale": 1.2, "noise": 0.2}
    _check_posterior(engine, kind, name, X, y, Xn, p)
    eps = rng.standard_normal((3, 2, M))
    th = {"k_length": np.tile(p["k_length"], (3, 1)) * rng.uniform(0.9, 1.1, (3, 2)), "k_scale": np.full(3, 1.2),
          "noise": np.array([0.2, 0.25, 0.3])}
    means, draws, infos = engine.predict_sweep(kind, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)
    assert np.all(infos == 0) and means.shape == (3, M) and draws.shape == (3, 2, M)
    for s in range(3):
        q = {"k_length": th["k_length"][s], "k_scale": 1.2, "noise": th["noise"][s]}
        m_ref, c_ref = ref.get_mvn_posterior(X, y, Xn, q, False, kernel=name, jitter=1e-6, route="inv")
        assert np.linalg.norm(means[s] - m_ref) <= 1e-8 * max(np.linalg.norm(m_ref), 1e-12)
        assert np.linalg.norm(draws[s] - ref.mvn_sample(m_ref, c_ref, eps[s])) <= 1e-7 * np.linalg.norm(draws[s])


def test_single_training_point_closed_form(engine):
    # N = 1 (the reference's y.squeeze() makes y 0-d there, gp.py:412; the device path has no such limit)
    x, yv, s2, noise, ell = 0.7, 1.3, 1.5, 0.2, 0.9
    Xn = np.array([[0.1], [0.7], [2.0]])
    engine.set_train(np.array([[x]]))
    lml, info = engine.factor(0, [ell], s2, noise, 1e-6, np.array([yv]))
    kxx = s2 + noise + 1e-6
    assert info == 0 and abs(lml - (-0.5 * yv * yv / kxx - 0.5 * np.log(kxx) - 0.5 * np.log(2 * np.pi))) < 1e-12
    mean, cov, var = engine.posterior(Xn, noise, 1e-6, want_cov=True, want_var=True)
    kp = s2 * np.exp(-0.5 * ((Xn[:, 0] - x) / ell) ** 2)
    np.testing.assert_allclose(mean, kp * yv / kxx, rtol=1e-12)
    kpp = s2 * np.exp(-0.5 * ((Xn[:, None, 0] - Xn[None, :, 0]) / ell) ** 2) + (noise + 1e-6) * np.eye(3)
    np.testing.assert_allclose(cov, kpp - np.outer(kp, kp) / kxx, rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(var, np.diag(cov), rtol=1e-11)


@pytest.mark.parametrize("kind,name", [(0, "RBF"), (1, "Matern"), (2, "Periodic")])
def test_maximum_input_dimension(engine, kind, name):
    N, M, d = 150, 40, 16  # GPX_MAX_DIM: the generic (non-templated) paths of Gram / k_pp / gradient
    rng = np.random.default_rng(7)
    X, Xn = rng.uniform(0, 1, (N, d)), rng.uniform(0, 1, (M, d))
    y = np.sin(X.sum(1))
    p = {"k_length": rng.uniform(1.0, 3.0, d), "k_scale": 1.1, "noise": 0.1}
    ell = p["k_length"]
    if kind == 2:
        p["period"] = 3.1
        ell = np.concatenate([ell, [3.1]])
    engine.set_train(X)
    lml, info = engine.factor(kind, ell, p["k_scale"], p["noise"], 1e-6, y)
    assert info == 0 and abs(lml - ref.exactgp_log_likelihood(X, y, p, kernel=name, jitter=1e-6)) <= 1e-10 * abs(lml)
    g_ell, g_scale, g_noise, alpha = engine.lml_grad()
    assert g_ell.shape == (d + (kind == 2),)
    if kind != 2:
        e_ell, e_scale, e_noise, _ = ref.exactgp_log_likelihood_grad(X, y, p, kernel=name, jitter=1e-6)
        sc = max(np.abs(e_ell).max(), abs(e_scale), abs(e_noise))
        np.testing.assert_allclose(g_ell, e_ell, rtol=1e-8, atol=1e-8 * sc)
    engine.factor(kind, ell, p["k_scale"], p["noise"], 1e-6, y)
    mean, cov, _ = engine.posterior(Xn, p["noise"], 1e-6)
    m_ref, c_ref = ref.get_mvn_posterior(X, y, Xn, p, False, kernel=name, jitter=1e-6, route="inv")
    assert np.linalg.norm(mean - m_ref) <= 1e-8 * np.linalg.norm(m_ref)
    assert np.linalg.norm(cov - c_ref) <= 1e-8 * np.linalg.norm(c_ref)
    with pytest.raises(RuntimeError):
        engine.set_train(rng.uniform(0, 1, (10, 17)))  # d > GPX_MAX_DIM is refused, not truncated
    engine.set_train(X)


def test_means_only_sweep_and_single_sample(engine):
    N, d, M = 90, 2, 21
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=4)
    th = bench_inputs.synthetic_theta_samples(1, d, seed=5)
    engine.set_train(X)
    means, draws, infos = engine.predict_sweep(1, th["k_length"], th["k_scale"], th["noise"], y, Xn, True, 1e-6, None)
    assert means.shape == (1, M) and draws.shape == (1, 0, M) and infos[0] == 0
    q = {"k_length": th["k_length"][0], "k_scale": th["k_scale"][0], "noise": th["noise"][0]}
    m_ref, _ = ref.get_mvn_posterior(X, y, Xn, q, True, kernel="Matern", jitter=1e-6, route="inv")
    assert np.linalg.norm(means[0] - m_ref) <= 1e-8 * np.linalg.norm(m_ref)


def test_degenerate_hyperparameters_give_nan_rows_not_crashes(engine):
    N, d, M, S = 200, 1, 30, 6
    rng = np.random.default_rng(0)
    X = np.sort(rng.uniform(0, 1, (N, 1)), axis=0)
    X[50] = X[49]  # duplicated input
    y = np.sin(6 * X[:, 0])
    Xn = rng.uniform(0, 1, (M, 1))
    ells = np.array([[0.3], [1e6], [1e-9], [0.3], [0.3], [0.3]])  # 1e6: K ~ rank one; 1e-9: K ~ diagonal
    scales = np.array([1.0, 1.0, 1.0, 1e-300, 1.0, 1.0])
    noises = np.array([0.05, 0.0, 0.0, 0.05, -1.0, np.nan])
    eps = rng.standard_normal((S, 1, M))
    engine.set_train(X)
    means, draws, infos = engine.predict_sweep(0, ells, scales, noises, y, Xn, False, 1e-6, eps)
    assert infos[0] == 0 and np.all(np.isfinite(means[0])) and np.all(np.isfinite(draws[0]))
    assert infos[2] == 0 and np.all(np.isfinite(means[2]))        # diagonal K + jitter is fine
    assert infos[4] != 0 and np.all(np.isnan(means[4])) and np.all(np.isnan(draws[4]))
    assert np.all(np.isnan(means[5])) or infos[5] != 0             # NaN noise poisons only its own row
    for s in (1, 3):  # numerically singular: either a clean failure (NaN row) or finite numbers, never garbage shapes
        assert (infos[s] != 0 and np.all(np.isnan(draws[s]))) or np.all(np.isfinite(means[s]))
    # the context is still usable afterwards
    lml, info = engine.factor(0, [0.3], 1.0, 0.05, 1e-6, y)
    assert info == 0 and np.isfinite(lml)


def test_bad_arguments_are_errors_not_crashes(engine):
    X, y, Xn, p = bench_inputs.synthetic_problem(20, 2, 5, seed=1)
    engine.set_train(X)
    with pytest.raises(NotImplementedError):
        from gpax_amd import _lib
        _lib.kernel_kind("NNGP")
    with pytest.raises(RuntimeError):
        engine.factor(7, p["k_length"], 1.0, 0.1, 1e-6, y)          # unknown kernel kind
    with pytest.raises(RuntimeError):
        engine.lml_grad()                                            # no factorisation to differentiate
    with pytest.raises(Exception):
        engine.predict_sweep(1, np.ones((2, 2)), np.ones(2), np.ones(2), y, Xn[:, :1], False, 1e-6, None)  # d mismatch
    lml, info = engine.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    assert info == 0


def test_lazy_far_updates_do_not_change_a_single_bit(monkeypatch):
    """linalg.hip potrf_lower: the bulk of the trailing matrix may take the panels of G outer blocks in one launch
    (K = 512 G).  With accumulators starting from -C every tile's updates form ONE fma chain however the k range is
    cut into launches, so the factor, the ride-along solve and the draws are bit-identical for every G (and for the
    outer blocking itself)."""
    import numpy as np
    from bench_inputs import synthetic_problem
    from gpax_amd import _lib

    N, d, M = 2700, 2, 300  # 22 diagonal blocks = 6 outer blocks: several groups, a ragged last one, ride-along rows
    X, y, Xn, p = synthetic_problem(N, d, M, seed=13)
    eps = np.random.default_rng(1).standard_normal((2, 1, M))
    ells = np.stack([p["k_length"], 1.1 * p["k_length"]])
    outs = []
    # (no switch: 22 tile rows run as ONE outer block since round 4 — common.h ONE_BLOCK_TILES; an explicit
    # GPX_OUTER_TILES brings back the blocked look-ahead schedule the lazy groups and the tail rule belong to)
    variants = [dict(), dict(GPX_OUTER_TILES="4"), dict(GPX_OUTER_TILES="4", GPX_TAIL_TILES="12", GPX_LAZY_GROUP="2"),
                dict(GPX_OUTER_TILES="4", GPX_TAIL_TILES="9"),
                dict(GPX_OUTER_TILES="4", GPX_LAZY_GROUP="3", GPX_TAIL_TILES="14"), dict(GPX_OUTER_TILES="4", GPX_LAZY_GROUP="1"),
                dict(GPX_OUTER_TILES="4", GPX_LAZY_GROUP="3"),
                dict(GPX_OUTER_TILES="4", GPX_LAZY_GROUP="4", GPX_TAIL_TILES="0"), dict(GPX_OUTER_TILES="2"),
                dict(GPX_LAZY_GROUP="2", GPX_OUTER_TILES="3"), dict(GPX_OUTER_TILES="8", GPX_LAZY_GROUP="1"),
                dict(GPX_OUTER_TILES="1"),
                # big-tile GEMMs of the sweeps as plain launches instead of persistent ones
                dict(GPX_PERSIST_SCOPE="0"),
                # k-step of the latency shapes, the diagonal-block kernels
                dict(GPX_SMALL_BK="32"), dict(GPX_SMALL_BK="16", GPX_LAZY_GROUP="3", GPX_OUTER_TILES="4"),
                # the round-1 register-staged latency-shape GEMM (the default since round 5: lat_tile, LDS-direct staging)
                dict(GPX_LAT_GEMM="r1"), dict(GPX_LAT_GEMM="r1", GPX_SMALL_BK="32", GPX_OUTER_TILES="2"),
                dict(GPX_LAT_GEMM="r5"), dict(GPX_LAT_GEMM="r5", GPX_OUTER_TILES="4"),
                dict(GPX_POTF2="tile", GPX_OUTER_TILES="2"),
                # lower-tile launches over the full square grid (round 5 default: only the live tiles are launched, and the
                # rank-128 update of a one-block chain keeps the 64 x 64 shape at every size)
                dict(GPX_LAT_LIN="0"), dict(GPX_LAT_LIN="0", GPX_PERSIST_SCOPE="0"), dict(GPX_LAT_LIN="0", GPX_LAT_GEMM="r1")]
    switches = ("GPX_LAZY_GROUP", "GPX_OUTER_TILES", "GPX_PERSIST_SCOPE", "GPX_TAIL_TILES", "GPX_SMALL_BK", "GPX_POTF2",
                "GPX_LAT_GEMM", "GPX_LAT_LIN")
    for env in variants:
        for k in switches:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = _lib.Engine(0)
        e.set_train(X)
        lml, info = e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        mean, cov, _ = e.posterior(Xn, p["noise"], 1e-6)
        sweep = e.predict_sweep(1, ells, [p["k_scale"]] * 2, [p["noise"]] * 2, y, Xn, False, 1e-6, eps)
        lml2, _ = e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        grad = e.lml_grad()
        outs.append((lml, mean, cov, sweep[0], sweep[1], grad[0], grad[3]))
        assert info == 0
        e.close()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.gpu
def test_the_two_diagonal_block_kernels_are_bit_identical(monkeypatch):
    """potf2_slim.h (default: memory-resident tiles, 94 VGPRs / 28 KB LDS, placed at once beside two resident
    trailing-update workgroups) and potf2_tile.h (round 2, four phases: the reference kept in the product)
    run tile for tile the same MFMA sequences: factors, block inverses (through the solves they feed), log-likelihood,
    gradient, posterior and the pivot report agree bit for bit — batched launches, a leading dimension that is not the
    block's own, and a matrix that is not positive definite included.  The kernel is switched in ONE context
    (gpx_debug_set_potf2), as bench.py's in-process A/B does."""
    import numpy as np
    from bench_inputs import synthetic_problem, synthetic_theta_samples
    from gpax_amd import _lib

    rng = np.random.default_rng(5)
    mats = []
    for n in (16, 100, 128, 200, 640, 1100):
        B = rng.standard_normal((n, n + 8))
        mats.append(B @ B.T / n + 0.3 * np.eye(n))
    bad = mats[3].copy()
    bad[150, 150] = -1.0  # pivot 151 fails
    mats.append(bad)
    worse = mats[2].copy()
    worse[0, 0] = 0.0  # the very first pivot fails: everything after it is NaN in both
    mats.append(worse)
    X, y, Xn, p = synthetic_problem(900, 2, 70, seed=3)
    th = synthetic_theta_samples(5, 2, seed=4)
    eps = np.random.default_rng(1).standard_normal((5, 1, 70))
    res = {}
    e = _lib.Engine(0)
    for mode in ("tile", "slim"):
        e.set_potf2(mode)
        out = [e.potrf(A) for A in mats]
        e.set_train(X)
        lml, info = e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        grad = e.lml_grad()
        sweep = e.predict_sweep(1, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)  # batched potf2
        res[mode] = (out, lml, info, grad, sweep)
    with pytest.raises(RuntimeError):
        e.set_potf2("column")
    with pytest.raises(RuntimeError):
        e.set_potf2("chain")  # round 3's register-resident kernel left the library in round 5
    e.close()
    for mode in ("slim",):
        for (L0, i0), (L1, i1) in zip(res["tile"][0], res[mode][0]):
            assert i0 == i1
            assert np.array_equal(L0, L1, equal_nan=True)
        assert res["tile"][1] == res[mode][1] and res["tile"][2] == res[mode][2] == 0
        for a, b in zip(res["tile"][3], res[mode][3]):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
        for a, b in zip(res["tile"][4], res[mode][4]):
            np.testing.assert_array_equal(a, b)
    assert [i for _, i in res["slim"][0]] == [0, 0, 0, 0, 0, 0, 151, 1]
    L = res["slim"][0][4][0]
    assert np.abs(np.tril(L) @ np.tril(L).T - mats[4]).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("N", [300, 1500, 2700])
def test_linv_t_tree_and_sweep_give_the_same_gradient(monkeypatch, N):
    """linalg.hip linv_t_tree (default) against the right-looking sweep (GPX_LINVT=sweep): two summation orders of the
    same L^-T — the lml gradient (gp.py:160-164 under autodiff in the reference) agrees to rounding, single and batched,
    at ragged sizes (N = 300: 3 tiles, a ragged pair at the first level; 2700: 22 tiles, ragged pairs at three levels)."""
    from gpax_amd import _lib
    X, y, _, p = bench_inputs.synthetic_problem(N, 2, 4, seed=17)
    B = 3
    ells = np.stack([p["k_length"] * f for f in (1.0, 1.2, 0.8)])
    scales, noises = np.full(B, p["k_scale"]), np.full(B, p["noise"])
    outs = {}
    for mode in ("tree", "sweep"):
        monkeypatch.setenv("GPX_LINVT", mode)
        e = _lib.Engine(0)
        e.set_train(X)
        e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        single = e.lml_grad()
        batched = e.fit_batch(1, ells, scales, noises, 1e-6, y)
        outs[mode] = (np.concatenate([np.ravel(single[0]), [single[1], single[2]]]), batched)
        e.close()
    g_tree, g_sweep = outs["tree"][0], outs["sweep"][0]
    np.testing.assert_allclose(g_tree, g_sweep, rtol=1e-9, atol=1e-9 * np.abs(g_sweep).max())
    g_ell, g_scale, g_noise, _ = ref.exactgp_log_likelihood_grad(X, y, p, kernel="Matern", jitter=1e-6)
    expect = np.concatenate([g_ell, [g_scale, g_noise]])
    np.testing.assert_allclose(g_tree, expect, rtol=1e-6, atol=1e-7 * np.abs(expect).max())
    for a, b in zip(outs["tree"][1], outs["sweep"][1]):
        np.testing.assert_allclose(np.asarray(a, dtype=float), np.asarray(b, dtype=float), rtol=1e-9, atol=1e-9)
    # entry 0 of the batch is the single-sample fit step, bit for bit
    np.testing.assert_array_equal(np.ravel(outs["tree"][1][2][0]), g_tree)


@pytest.mark.gpu
def test_live_tile_grids_give_the_bits_of_the_square_grids_in_batched_sweeps(monkeypatch):
    """gemm_f64.hip: a lower-tile launch enumerates only its live tiles (round 5; GPX_LAT_LIN=0: the square grid whose upper
    half returns at once) — with the batch in grid.z and the entries dealt to the XCDs in groups of eight (batch_xcd_order),
    with ride-along rows below the square part, with a ragged last group.  A tile's arithmetic does not depend on where in the
    grid it sits: factor, gradient, posterior and draws of a 19-sample sweep agree bit for bit."""
    import numpy as np
    from bench_inputs import synthetic_problem, synthetic_theta_samples
    from gpax_amd import _lib

    N, d, M, S = 1300, 2, 200, 19
    X, y, Xn, _ = synthetic_problem(N, d, M, seed=5)
    th = synthetic_theta_samples(S, d, seed=6)
    eps = np.random.default_rng(7).standard_normal((S, 2, M))
    outs = []
    for lin in ("1", "0"):
        monkeypatch.setenv("GPX_LAT_LIN", lin)
        e = _lib.Engine(0)
        e.set_train(X)
        sweep = e.predict_sweep(1, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)
        fb = e.fit_batch(1, th["k_length"], th["k_scale"], th["noise"], 1e-6, y)
        e.close()
        outs.append((sweep[0], sweep[1], fb[0], fb[1], fb[2], fb[3]))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    assert np.all(np.isfinite(outs[0][0])) and np.all(outs[0][3] == 0)


@pytest.mark.gpu
def test_panel_trsm_riding_in_the_potf2_launch_gives_the_same_bits():
    """potf2.hip potf2_trsm_kernel (round 6): in one-outer-block factorisations of a single sample the panel TRSM of a
    step is done by workgroups of the potf2 launch that wait for a device flag, instead of a launch of its own
    (gp.py:160-164's Cholesky, 3 -> 2 launches per 128 columns).  Same strip body, same bits: factors (padded orders, a
    failing pivot in a block whose strips are already waiting), lml, gradient, the posterior with the k_pX rows riding along,
    switched in ONE context, several times over (the flag's epoch keeps counting)."""
    import numpy as np
    from bench_inputs import synthetic_problem
    from gpax_amd import _lib

    rng = np.random.default_rng(7)
    mats = []
    for n in (129, 300, 1100, 2048):
        B = rng.standard_normal((n, n + 8))
        mats.append(B @ B.T / n + 0.3 * np.eye(n))
    bad = mats[2].copy()
    bad[150, 150] = -1.0  # pivot 151 fails in the second block: NaN from there on, every later flag still published
    mats.append(bad)
    X, y, Xn, p = synthetic_problem(1900, 2, 200, seed=3)
    e = _lib.Engine(0)
    res = {}
    for rnd, mode in enumerate(("nofuse", "fuse", "nofuse", "fuse")):
        e.set_potf2(mode)
        out = [e.potrf(A) for A in mats]
        e.set_train(X)
        lml, info = e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        grad = e.lml_grad()
        e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)  # (the gradient pass left K^-1 in place of L)
        post = e.posterior(Xn, p["noise"], 1e-6, want_cov=True, want_var=True)
        res[rnd] = (out, lml, info, grad, post)
    # one order, many different matrices back to back: the strips read L^-1 of a block through their XCD's L2 without an
    # invalidate of their own (potf2.hip) — a line left over from the factorisation before would show here
    soak = [(lambda B: B @ B.T / 1024 + (0.2 + 0.05 * k) * np.eye(1024))(rng.standard_normal((1024, 1032))) for k in range(12)]
    e.set_potf2("nofuse")
    want = [e.potrf(A)[0] for A in soak]
    e.set_potf2("fuse")
    for rep in range(3):
        for A, L in zip(soak, want):
            assert np.array_equal(e.potrf(A)[0], L)
    e.close()
    for rnd in (1, 2, 3):
        for (L0, i0), (L1, i1) in zip(res[0][0], res[rnd][0]):
            assert i0 == i1
            assert np.array_equal(L0, L1, equal_nan=True)
        assert res[0][1] == res[rnd][1] and res[0][2] == res[rnd][2] == 0
        for a, b in zip(res[0][3], res[rnd][3]):
            np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
        for a, b in zip(res[0][4], res[rnd][4]):
            np.testing.assert_array_equal(a, b)
    assert [i for _, i in res[1][0]] == [0, 0, 0, 0, 151]


@pytest.mark.gpu
def test_fused_chain_steps_of_several_contexts_running_at_once_do_not_wait_for_each_other_forever():
    """potf2_trsm_kernel's strip workgroups spin on a flag and hold a whole CU each (128 KB of LDS): three contexts factoring
    at once ask for more spinning workgroups than the chip has CUs.  Every strip waits only for workgroup 0 of its OWN launch,
    which is dispatched before it, so the earliest launch always finishes — no deadlock, and the same bits as one context
    alone.  (Three host threads, one context each, 4096 / 2048 / 3000-row factorisations interleaved.)"""
    import threading

    import numpy as np
    from gpax_amd import _lib

    rng = np.random.default_rng(13)
    mats = []
    for n in (4096, 2048, 3000):
        B = rng.standard_normal((n, n + 8))
        mats.append(B @ B.T / n + 0.3 * np.eye(n))
    ref_eng = _lib.Engine(0)
    want = [ref_eng.potrf(A)[0] for A in mats]
    ref_eng.close()
    engines = [_lib.Engine(0) for _ in range(3)]
    bad, errs = [], []

    def work(t):
        try:
            for rep in range(6):
                k = (t + rep) % 3
                L, info = engines[t].potrf(mats[k])
                if info != 0 or not np.array_equal(L, want[k]):
                    bad.append((t, rep, k))
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a factorisation never returned"
    for e in engines:
        e.close()
    assert not errs and not bad, (errs, bad)
