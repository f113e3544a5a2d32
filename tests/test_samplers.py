"""CPU: the host samplers and prior distributions in isolation."""
import numpy as np
import pytest
import scipy.stats as st

from gpax_amd.infer import dist
from gpax_amd.infer.nuts import adaptation_schedule, run_nuts
from gpax_amd.infer.svi import Adam, fit_delta, fit_normal


@pytest.mark.parametrize("d,ref", [
    (dist.Normal(0.3, 1.7), st.norm(0.3, 1.7)),
    (dist.LogNormal(0.2, 0.6), st.lognorm(s=0.6, scale=np.exp(0.2))),
    (dist.HalfNormal(0.8), st.halfnorm(scale=0.8)),
    (dist.Gamma(2.0, 5.0), st.gamma(a=2.0, scale=1 / 5.0)),
    (dist.Uniform(1.0, 3.0), st.uniform(1.0, 2.0)),
])
def test_distributions_against_scipy(d, ref):
    x = np.array([1.2, 1.9, 2.5])
    np.testing.assert_allclose(d.log_prob(x), ref.logpdf(x), rtol=1e-12)
    h = 1e-6
    np.testing.assert_allclose(d.grad_log_prob(x), (ref.logpdf(x + h) - ref.logpdf(x - h)) / (2 * h), rtol=1e-5,
                               atol=1e-7)
    assert abs(d.median() - ref.median()) < 1e-9
    u = np.array([-0.5, 0.1, 0.9])
    xx = d.transform(u)
    np.testing.assert_allclose(d.inverse(xx), u, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(d.dx_du(u), (d.transform(u + h) - d.transform(u - h)) / (2 * h), rtol=1e-6)
    lj, dlj = d.log_abs_det_jacobian(u)
    np.testing.assert_allclose(lj, np.log(np.abs(d.dx_du(u))), rtol=1e-10, atol=1e-12)
    s = d.sample(np.random.default_rng(0), (4000,))
    assert abs(np.median(s) - ref.median()) < 0.1


def test_adaptation_schedule_covers_warmup():
    for n in [10, 50, 150, 1000, 2000]:
        sched = adaptation_schedule(n)
        assert sched[0][0] == 0 and sched[-1][1] == n - 1
        for (a, b), (c, d) in zip(sched[:-1], sched[1:]):
            assert c == b + 1


def test_nuts_recovers_correlated_gaussian():
    A = np.array([[2.0, 0.9, 0.0], [0.9, 1.0, 0.2], [0.0, 0.2, 0.5]])
    P = np.linalg.inv(A)
    res = run_nuts(lambda u: (0.5 * u @ P @ u, P @ u), np.ones(3), 400, 3000, np.random.default_rng(0))
    d = res["draws"]
    assert np.abs(d.mean(0)).max() < 0.15
    np.testing.assert_allclose(np.cov(d.T), A, atol=0.25)
    assert res["diverging"].sum() == 0 and 0.6 < res["accept"].mean() <= 1.0


def test_nuts_handles_infinite_potential_regions():
    # half-line: U = u^2/2 for u > -1, +inf otherwise
    def pot(u):
        if u[0] <= -1:
            return np.inf, np.zeros(1)
        return 0.5 * u[0] ** 2, u.copy()
    res = run_nuts(pot, np.array([0.5]), 200, 1000, np.random.default_rng(1))
    assert res["draws"].min() > -1


def test_adam_and_guides_on_a_quadratic():
    target = np.array([1.0, -2.0])
    obj = lambda u: (-0.5 * np.sum((u - target) ** 2), -(u - target))
    u, losses = fit_delta(obj, np.zeros(2), 800, 0.05)
    np.testing.assert_allclose(u, target, atol=1e-2)
    assert losses[-1] < losses[0]
    loc, scale, _ = fit_normal(obj, 2, 3000, 0.02, np.random.default_rng(0))
    np.testing.assert_allclose(loc, target, atol=0.2)
    np.testing.assert_allclose(scale, 1.0, atol=0.3)
    a = Adam(2, 0.1)
    x = a.step(np.zeros(2), np.array([1.0, -1.0]))
    np.testing.assert_allclose(x, [-0.1, 0.1], rtol=1e-6)
