"""Known-answer tests for the host-side samplers (NumPyro's NUTS / SVI restated in
gpax_amd/infer).  Chains cannot be compared with a NumPyro run here (no jax), so next to the distributional
tests of tests/test_samplers.py every deterministic building block is pinned to the values its published
definition gives: Stan's warm-up windows (reference manual, "Automatic parameter tuning": initial buffer 75,
base window 25 doubling, terminal buffer 50 — numpyro.infer.hmc_util.build_adaptation_schedule follows it),
the dual-averaging recurrence of Hoffman & Gelman 2014 (Alg. 5 / 6: gamma 0.05, t0 10, kappa 0.75,
mu = log(10 eps0)), the leapfrog map on a harmonic oscillator, the U-turn rule on an exactly periodic orbit,
Stan's regularised Welford variance and Kingma & Ba's Adam with the b1 = 0.5 gpax passes (vigp.py:108)."""
import math

import numpy as np

from gpax_amd.infer.nuts import (_DualAveraging, _Welford, _leapfrog, adaptation_schedule, nuts_transition)
from gpax_amd.infer.svi import Adam


def test_warmup_windows_are_stans_published_schedule():
    assert adaptation_schedule(1000) == [(0, 74), (75, 99), (100, 149), (150, 249), (250, 449), (450, 949), (950, 999)]
    # 2000 (gpax's default num_warmup, gp.py:171): 75 | 25 50 100 200 400 then the rest up to the 50-step tail
    assert adaptation_schedule(2000) == [(0, 74), (75, 99), (100, 149), (150, 249), (250, 449), (450, 849), (850, 1949),
                                         (1950, 1999)]
    # short warm-ups: 15 % / 75 % / 10 % split, below 20 steps a single window
    assert adaptation_schedule(100) == [(0, 14), (15, 89), (90, 99)]
    assert adaptation_schedule(19) == [(0, 18)]


def test_dual_averaging_follows_hoffman_gelman():
    eps0, delta = 0.5, 0.8
    da = _DualAveraging(eps0, delta)
    mu = math.log(10 * eps0)
    # step 1, accept 0.3: Hbar = (delta - a) / (1 + t0); log eps = mu - sqrt(1) / gamma * Hbar; averaged = log eps
    da.update(0.3)
    h1 = (0.8 - 0.3) / 11.0
    assert math.isclose(da.h_bar, h1, rel_tol=1e-15)
    assert math.isclose(da.log_eps, mu - h1 / 0.05, rel_tol=1e-15)
    assert math.isclose(da.log_eps_bar, da.log_eps, rel_tol=1e-15)
    # step 2, accept 1.0
    le1 = da.log_eps
    da.update(1.0)
    h2 = (1 - 1 / 12.0) * h1 + (1 / 12.0) * (0.8 - 1.0)
    le2 = mu - math.sqrt(2.0) / 0.05 * h2
    eta = 2.0 ** -0.75
    assert math.isclose(da.h_bar, h2, rel_tol=1e-15)
    assert math.isclose(da.log_eps, le2, rel_tol=1e-15)
    assert math.isclose(da.log_eps_bar, eta * le2 + (1 - eta) * le1, rel_tol=1e-15)
    # a chain that always accepts at the target leaves the step at mu = log(10 eps0)
    da = _DualAveraging(eps0, delta)
    for _ in range(50):
        da.update(delta)
    assert math.isclose(da.log_eps, mu, abs_tol=1e-15)


def test_leapfrog_is_the_stoermer_verlet_map():
    pe = lambda u: (0.5 * float(u @ u), u.copy())  # unit harmonic oscillator
    u, p, eps = np.array([0.7, -0.2]), np.array([0.3, 1.1]), 0.25
    U, g = pe(u)
    u1, p1, U1, g1 = _leapfrog(pe, u, p, g, eps, np.ones(2))
    np.testing.assert_allclose(u1, u * (1 - eps ** 2 / 2) + eps * p, rtol=1e-15)
    np.testing.assert_allclose(p1, p * (1 - eps ** 2 / 2) - eps * u * (1 - eps ** 2 / 4), rtol=1e-15)
    assert U1 == 0.5 * float(u1 @ u1) and np.array_equal(g1, u1)
    # time reversal: negate the momentum, step, negate again -> the starting point
    u0, p0, _, _ = _leapfrog(pe, u1, -p1, g1, eps, np.ones(2))
    np.testing.assert_allclose(u0, u, atol=1e-15)
    np.testing.assert_allclose(-p0, p, atol=1e-15)
    # mass matrix: u advances with M^-1 p
    inv_mass = np.array([4.0, 0.25])
    u2, _, _, _ = _leapfrog(pe, u, p, g, eps, inv_mass)
    np.testing.assert_allclose(u2, u + eps * inv_mass * (p - 0.5 * eps * u), rtol=1e-15)


def test_tree_doubling_never_runs_past_half_a_period_of_a_harmonic_orbit():
    """1-D unit oscillator: a trajectory that spans more than half a period (pi) has reversed its momentum, so the
    generalised U-turn rule (sum of momenta against the end momenta, numpyro.infer.hmc_util._is_turning) must have
    fired: with eps = 0.12 no transition may take a sixth doubling (32 points span 3.72 > pi), whatever the random
    directions and the momentum draw; and the exact orbit keeps the energy error of every proposal tiny."""
    pe = lambda u: (0.5 * float(u @ u), u.copy())
    seen = set()
    for seed in range(200):
        rng = np.random.default_rng(seed)
        u = np.array([rng.standard_normal()])
        U, g = pe(u)
        _, _, _, acc, n, div = nuts_transition(pe, u, U, g, 0.12, np.ones(1), rng)
        assert 1 <= n <= 31 and not div and 0.99 < acc <= 1.0
        seen.add(n)
    assert 31 in seen and 1 in seen  # some orbits need all five doublings, some turn at once
    # max_tree_depth caps the trajectory at 2^depth - 1 leapfrogs when nothing turns
    rng = np.random.default_rng(0)
    _, _, _, _, n, _ = nuts_transition(pe, np.array([0.1]), *pe(np.array([0.1])), 1e-3, np.ones(1), rng, max_tree_depth=4)
    assert n == 15


def test_uturn_rule_is_the_generalised_criterion():
    from gpax_amd.infer.nuts import _uturn
    inv_mass = np.array([1.0, 4.0])
    pl, pr, rho = np.array([0.2, 0.1]), np.array([1.0, 0.0]), np.array([3.0, 0.5])
    r = rho - 0.5 * (pl + pr)  # the end points count one half each
    assert not _uturn(rho, pl, pr, inv_mass) and r @ (inv_mass * pl) > 0 and r @ (inv_mass * pr) > 0
    pr2 = np.array([-1.0, 0.9])  # r = (2.4, 0) against M^-1 pr2 = (-1, 3.6): -2.4 <= 0
    assert _uturn(rho, pl, pr2, inv_mass)
    pl2 = np.array([-3.0, 0.1])
    assert _uturn(rho, pl2, pr, inv_mass)
    # a two-point trajectory (rho = p_left + p_right): r = (p_left + p_right) / 2, turning iff the momenta oppose
    assert not _uturn(np.array([2.0]), np.array([1.0]), np.array([1.0]), np.ones(1))
    assert _uturn(np.array([0.5]), np.array([1.0]), np.array([-0.5]), np.ones(1))
    assert _uturn(np.array([0.0]), np.array([1.0]), np.array([-1.0]), np.ones(1))  # exactly zero counts as turning


def test_welford_variance_carries_stans_regularisation():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((40, 3)) * np.array([0.1, 1.0, 10.0])
    wf = _Welford(3)
    for row in x:
        wf.update(row)
    n = 40
    np.testing.assert_allclose(wf.mean, x.mean(0), rtol=1e-13)
    np.testing.assert_allclose(wf.variance(), (n / (n + 5.0)) * x.var(0, ddof=1) + 1e-3 * (5.0 / (n + 5.0)), rtol=1e-13)


def test_adam_two_steps_by_hand_with_gpax_b1():
    lr, b1, b2, e = 0.1, 0.5, 0.999, 1e-8
    a = Adam(1, lr, b1=b1)
    g1, g2 = 2.0, -1.0
    x1 = a.step(np.array([0.0]), np.array([g1]))
    m1, v1 = (1 - b1) * g1, (1 - b2) * g1 * g1
    want1 = -lr * (m1 / (1 - b1)) / (math.sqrt(v1 / (1 - b2)) + e)
    np.testing.assert_allclose(x1, [want1], rtol=1e-14)
    x2 = a.step(x1, np.array([g2]))
    m2, v2 = b1 * m1 + (1 - b1) * g2, b2 * v1 + (1 - b2) * g2 * g2
    want2 = want1 - lr * (m2 / (1 - b1 ** 2)) / (math.sqrt(v2 / (1 - b2 ** 2)) + e)
    np.testing.assert_allclose(x2, [want2], rtol=1e-14)
