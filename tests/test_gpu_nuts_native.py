"""GPU: the NUTS transition inside the library (gpx_nuts_transition, csrc/nuts.hip) against the Python loop it restates
(gpax_amd/infer/nuts.py; numpyro.infer.NUTS under ExactGP.fit, gpax/models/gp.py:207-218).  Both draw from the same PCG64
stream in the same order and evaluate the same device potential, so they build the SAME trees; the floating-point
operations of the host arithmetic may round the last bit differently (NumPy's exp / log / dot vs libm), hence: tree sizes,
divergence flags and generator states identical, positions equal to 1e-9 per transition — and bit for bit once the Python
loop is given the library's scalar arithmetic (its potential, dot products summed in order)."""
import numpy as np
import pytest

import bench_inputs
from gpax_amd import _lib
from gpax_amd.infer.nuts import nuts_transition
from gpax_amd.models import ExactGP
from gpax_amd.utils import get_keys

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def real_engine(engine):
    _lib.set_engine(engine)
    yield
    _lib.set_engine(None)


def _model(kernel, N, d, seed):
    X, y, _, _ = bench_inputs.synthetic_problem(N, d, 4, seed=seed)
    m = ExactGP(d, kernel)
    m.X_train, m.y_train = m._set_data(X, y)
    m._data_version += 1
    return m


@pytest.mark.parametrize("kernel,N,d", [("RBF", 25, 1), ("Matern", 90, 2), ("Periodic", 40, 1), ("RBF", 200, 3)])
def test_native_transition_builds_the_tree_the_python_loop_builds(kernel, N, d):
    m = _model(kernel, N, d, seed=N)
    sites = m._sites()
    dim = sum(s.size for s in sites)

    def potential(u):
        v, g = m._log_joint(sites, u, 1e-6, jacobian=True)
        return (-v, -g) if np.isfinite(v) else (np.inf, np.zeros_like(u))

    rng_a, rng_b = np.random.default_rng(11), np.random.default_rng(11)
    native = m._native_transition(sites, 1e-6, rng_b)
    assert native is not None
    u = np.log(np.full(dim, 0.7)) + 0.1 * np.arange(dim)
    U, g = potential(u)
    ua, Ua, ga = u.copy(), U, g.copy()
    ub, Ub, gb = u.copy(), U, g.copy()
    inv_mass = np.linspace(0.5, 1.5, dim)
    sizes = []
    for it in range(25):
        eps = [0.05, 0.2, 0.6, 1.5, 6.0][it % 5]  # short and long trees, and steps that diverge
        ua, Ua, ga, acc_a, nl_a, div_a = nuts_transition(potential, ua, Ua, ga, eps, inv_mass, rng_a, 6)
        ub, Ub, gb, acc_b, nl_b, div_b = native(ub, Ub, gb, eps, inv_mass, rng_b, 6)
        assert (nl_a, div_a) == (nl_b, div_b), (it, nl_a, nl_b)
        assert rng_a.bit_generator.state == rng_b.bit_generator.state  # the same number of uniforms and normals consumed
        np.testing.assert_allclose(ub, ua, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(gb, ga, rtol=1e-7, atol=1e-9)
        assert abs(Ub - Ua) <= 1e-9 * max(1.0, abs(Ua)) and abs(acc_a - acc_b) < 1e-9
        sizes.append(nl_a)
    assert max(sizes) >= 15 and min(sizes) <= 3  # the cases exercised trees of several depths


def test_the_two_loops_are_the_same_algorithm_bit_for_bit_when_they_share_the_scalar_arithmetic(monkeypatch):
    """The Python loop, unmodified in its control flow, driven with the library's potential (gpx_nuts_potential) and with its
    two dot products summed in order instead of through BLAS: every statement of the two loops is then the same IEEE
    operation, and 200 transitions of mixed step sizes — trees of depth 0 to 8, divergences — end in the same bits."""
    from gpax_amd.infer import nuts as pynuts
    m = _model("Matern", 60, 2, seed=7)
    sites = m._sites()
    dim = sum(s.size for s in sites)
    rng_a, rng_b = np.random.default_rng(3), np.random.default_rng(3)
    native = m._native_transition(sites, 1e-6, rng_b)
    eng = m._engine()

    def potential(u):
        return eng.nuts_potential(native.plan, u)

    def wsum(a, inv_mass, b):
        return float(np.sum(a * (inv_mass * b)))

    def energy(U, p, inv_mass):
        return np.inf if not np.isfinite(U) else U + 0.5 * wsum(p, inv_mass, p)

    def uturn(rho, pl, pr, inv_mass):
        rho = rho - 0.5 * (pl + pr)
        return (wsum(rho, inv_mass, pl) <= 0) or (wsum(rho, inv_mass, pr) <= 0)

    monkeypatch.setattr(pynuts, "_energy", energy)
    monkeypatch.setattr(pynuts, "_uturn", uturn)
    u = np.log(np.array([1.1, 0.8, 1.3, 0.2]))[:dim]
    U, g = potential(u)
    ua, Ua, ga = u.copy(), U, g.copy()
    ub, Ub, gb = u.copy(), U, g.copy()
    inv_mass = np.linspace(0.6, 1.4, dim)
    depths = set()
    for it in range(200):
        eps = [0.02, 0.1, 0.3, 0.8, 2.5, 9.0][it % 6]
        ua, Ua, ga, acc_a, nl_a, div_a = pynuts.nuts_transition(potential, ua, Ua, ga, eps, inv_mass, rng_a, 8)
        ub, Ub, gb, acc_b, nl_b, div_b = native(ub, Ub, gb, eps, inv_mass, rng_b, 8)
        assert (nl_a, div_a, acc_a, Ua) == (nl_b, div_b, acc_b, Ub), it
        np.testing.assert_array_equal(ua, ub)
        np.testing.assert_array_equal(ga, gb)
        depths.add(int(np.log2(nl_a + 1)))
    assert rng_a.bit_generator.state == rng_b.bit_generator.state
    assert len(depths) >= 5


def test_fit_draws_the_same_chain_with_and_without_the_native_loop(monkeypatch):
    X, y, _, _ = bench_inputs.synthetic_problem(25, 1, 4, seed=1)
    out = {}
    for native in ("0", "1"):
        monkeypatch.setenv("GPX_NATIVE_NUTS", native)
        m = ExactGP(1, "RBF")
        m.fit(get_keys()[0], X, y, num_warmup=60, num_samples=60, progress_bar=False, print_summary=False)
        st = m.mcmc.get_extra_fields()[0]
        out[native] = (m.get_samples(), st["n_leapfrog"].copy(), st["diverging"].copy(), float(st["step_size"]))
    # the same trees all along; positions agree to what 120 transitions of a chaotic map make of last-bit differences in
    # NumPy's exp / log / dot against libm's (csrc/nuts.hip header)
    np.testing.assert_array_equal(out["0"][1], out["1"][1])
    np.testing.assert_array_equal(out["0"][2], out["1"][2])
    assert abs(out["0"][3] - out["1"][3]) <= 1e-3 * out["0"][3]
    for k in out["0"][0]:
        np.testing.assert_allclose(out["1"][0][k], out["0"][0][k], rtol=1e-3)


def test_models_the_native_loop_does_not_cover_keep_the_python_loop():
    from gpax_amd import dist
    m = _model("RBF", 20, 1, seed=3)
    rng = np.random.default_rng(0)
    assert m._native_transition(m._sites(), 1e-6, rng) is not None
    assert ExactGP(1, "RBF", noise_prior_dist=dist.HalfNormal(0.5))._native_transition(
        ExactGP(1, "RBF", noise_prior_dist=dist.HalfNormal(0.5))._sites(), 1e-6, rng) is None
    mf = ExactGP(1, "RBF", mean_fn=lambda x: 0.1 * x[:, 0])
    mf.X_train, mf.y_train = m.X_train, m.y_train
    assert mf._native_transition(mf._sites(), 1e-6, rng) is None
    assert m._native_transition(m._sites(), 1e-6, np.random.Generator(np.random.MT19937(1))) is None
