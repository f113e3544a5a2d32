"""CPU: acquisition functions (SURVEY §8f "next" row 1) against scipy.stats closed forms and the reference's
own test expectations (gpax/tests/test_acq.py: shapes, penalties, argument validation)."""
import numpy as np
import pytest
from scipy.stats import norm

from gpax_amd import _lib
from gpax_amd.acquisition import EI, POI, UCB, UE, Thompson, compute_penalty, ei, poi, ucb, ue
from gpax_amd.models import ExactGP, viGP
from gpax_amd.utils import get_keys
from oracle import cpu_ref as ref
import bench_inputs
from tests.oracle_engine import OracleEngine


@pytest.fixture(autouse=True)
def oracle_engine():
    _lib.set_engine(OracleEngine())
    yield
    _lib.set_engine(None)


def test_base_functions_closed_forms():
    rng = np.random.default_rng(0)
    mean, var = rng.standard_normal(50), rng.uniform(0.1, 2.0, 50)
    s = np.sqrt(var)
    for maximize in [False, True]:
        best = mean.max() if maximize else mean.min()
        u = (mean - best) / s * (1 if maximize else -1)
        np.testing.assert_allclose(ei((mean, var), maximize=maximize), s * (norm.pdf(u) + u * norm.cdf(u)), rtol=1e-12)
        u2 = (mean - best - 0.01) / s * (1 if maximize else -1)
        np.testing.assert_allclose(poi((mean, var), maximize=maximize), norm.cdf(u2), rtol=1e-12)
    np.testing.assert_allclose(ucb((mean, var), beta=0.5, maximize=True), mean + np.sqrt(0.5 * var))
    np.testing.assert_allclose(ucb((mean, var), beta=0.5, maximize=False), -(mean - np.sqrt(0.5 * var)))
    np.testing.assert_allclose(ue((mean, var)), s)
    np.testing.assert_allclose(ei((mean, var), best_f=0.3), ei((mean, var), best_f=0.3, maximize=False))


def test_penalties():
    X = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 2.0]])
    recent = np.array([[1.0, 0.0]])
    np.testing.assert_array_equal(compute_penalty(X, recent, "delta"), [0.0, np.inf, 0.0])
    p = compute_penalty(X, recent, "inverse_distance", 2.0)
    np.testing.assert_allclose(p, 2.0 / (np.linalg.norm(X - recent, axis=1) + 1))
    recent2 = np.array([[0.0, 0.0], [2.0, 2.0]])  # [older, newer]: ages 3, 2
    p2 = compute_penalty(X, recent2, "inverse_distance")
    d = np.stack([np.linalg.norm(X - r, axis=1) for r in recent2], 1)
    np.testing.assert_allclose(p2, (1 / (d + 1) / np.array([3, 2])).sum(1))
    with pytest.raises(NotImplementedError):
        compute_penalty(X, recent, "bogus")


@pytest.mark.parametrize("acq", [EI, UCB, POI, UE])
def test_wrappers_with_mcmc_and_vi_models(acq):
    X, y, Xn, _ = bench_inputs.synthetic_problem(20, 1, 9, seed=2)
    m = ExactGP(1, "RBF")
    m.fit(get_keys()[0], X, y, num_warmup=15, num_samples=15, progress_bar=False, print_summary=False)
    a = acq(get_keys()[1], m, Xn[:, 0], n=2)
    assert a.shape == (9,) and np.isfinite(a).all()
    v = viGP(1, "Matern")
    v.fit(get_keys()[0], X, y, num_steps=20, progress_bar=False, print_summary=False)
    a2 = acq(get_keys()[1], v, Xn)
    assert a2.shape == (9,) and np.isfinite(a2).all()
    mean, var = v.predict(get_keys()[1], Xn)
    if acq is UE:
        np.testing.assert_allclose(a2, np.sqrt(var))
    with pytest.raises(ValueError):
        acq(get_keys()[1], v, Xn, penalty="delta")
    pen = acq(get_keys()[1], v, Xn, penalty="delta", recent_points=Xn[2:3])
    assert np.isneginf(pen[2]) and np.all(np.isfinite(np.delete(pen, 2)))
    pen2 = acq(get_keys()[1], v, Xn, penalty="inverse_distance", recent_points=Xn[2:4], penalty_factor=0.5)
    assert np.all(pen2 < a2)


def test_thompson_shapes():
    X, y, Xn, _ = bench_inputs.synthetic_problem(20, 1, 9, seed=2)
    m = ExactGP(1, "RBF")
    m.fit(get_keys()[0], X, y, num_warmup=15, num_samples=15, progress_bar=False, print_summary=False)
    t = Thompson(get_keys()[1], m, Xn)
    assert np.asarray(t).reshape(-1).shape == (9,)
    v = viGP(1, "RBF")
    v.fit(get_keys()[0], X, y, num_steps=10, progress_bar=False, print_summary=False)
    assert Thompson(get_keys()[1], v, Xn, noiseless=False).shape == (1, 9)


def test_base_functions_match_the_oracle_restatement_of_base_acq():
    """oracle/cpu_ref.py restates gpax/acquisition/base_acq.py:20-155 line by line (erf-based NumPyro Normal cdf);
    the product's closed forms (scipy ndtr) must agree element-wise, incl. best_f given / derived and both senses."""
    rng = np.random.default_rng(4)
    mean, var = 3.0 * rng.standard_normal(200), rng.uniform(1e-6, 4.0, 200)
    for maximize in (False, True):
        for best_f in (None, 0.7):
            # far in the tail sigma (phi(u) + u Phi(u)) cancels to ~1e-16 of its terms: absolute floor at 1e-14 of the scale
            e = ref.acq_ei((mean, var), best_f, maximize)
            np.testing.assert_allclose(ei((mean, var), best_f=best_f, maximize=maximize), e, rtol=1e-10,
                                       atol=1e-14 * np.abs(e).max())
            np.testing.assert_allclose(poi((mean, var), best_f=best_f, xi=0.03, maximize=maximize),
                                       ref.acq_poi((mean, var), best_f, 0.03, maximize), rtol=1e-10, atol=1e-15)
        np.testing.assert_allclose(ucb((mean, var), beta=0.4, maximize=maximize), ref.acq_ucb((mean, var), 0.4, maximize),
                                   rtol=1e-15)
    np.testing.assert_allclose(ue((mean, var)), ref.acq_ue((mean, var)), rtol=1e-15)


def test_readme_bayesian_optimisation_loop_finds_the_minimum():
    """examples/bo_loop.py = steps A - D of the reference README (fit, UCB on the unmeasured grid, argmax, measure),
    here on the test-only oracle engine: the loop localises the global minimum of the 1-D black box."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import bo_loop
    out = bo_loop.main(num_steps=8, num_warmup=40, num_samples=40, verbose=False)
    assert out["n_measured"] == 12
    assert abs(out["x_best"] - out["x_true"]) <= 0.1 and out["y_best"] <= out["y_true"] + 0.1


def test_penalties_reproduce_the_known_answers_of_the_reference_tests():
    """The only literal expected values the reference's test-suite holds for this row (gpax tests/test_acq.py:232-247)."""
    from gpax_amd.acquisition.penalties import compute_penalty
    X = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=float)
    pen = compute_penalty(X, np.array([[4, 5, 6], [1, 2, 3]], dtype=float), "delta", 1.0)
    np.testing.assert_array_equal(pen, [np.inf, np.inf, 0.0])
    pen = compute_penalty(X, np.array([[4, 5, 6], [7, 8, 9]], dtype=float), "inverse_distance", 1.0)
    assert pen[-1] > pen[-2] > pen[-3]
