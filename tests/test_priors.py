"""gpax_amd.priors / utils.set_fn — mirrors of gpax/tests/test_priors.py (dist helpers, auto priors) and
test_func_setter.py::test_set_fn."""
import numpy as np
import pytest

from gpax_amd import dist, priors
from gpax_amd.priors import (auto_lognormal_priors, auto_normal_priors, auto_priors, gamma_dist, halfnormal_dist,
                             lognormal_dist, normal_dist, uniform_dist)
from gpax_amd.utils import set_fn


def sample_function(x, a, b):
    return a + b * x


def test_get_uniform_dist():  # test_priors.py:78-102
    u = uniform_dist(low=1.0, high=5.0)
    assert isinstance(u, dist.Uniform) and u._bounds() == (1.0, 5.0)
    u = uniform_dist(input_vec=np.array([1.0, 2.0, 3.0, 4.0, 5.0]))
    assert u._bounds() == (1.0, 5.0)
    u = uniform_dist(low=2.0, input_vec=np.array([1.0, 5.0]))
    assert u._bounds() == (2.0, 5.0)


def test_get_gamma_normal_lognormal_halfnormal_dist():  # test_priors.py:105-138
    g = gamma_dist(c=2.0, r=1.0)
    assert isinstance(g, dist.Gamma) and (g.concentration, g.rate) == (2.0, 1.0)
    n = normal_dist(loc=2.0, scale=3.0)
    assert isinstance(n, dist.Normal) and (n.loc, n.scale) == (2.0, 3.0)
    ln = lognormal_dist(loc=2.0, scale=3.0)
    assert isinstance(ln, dist.LogNormal) and (ln.loc, ln.scale) == (2.0, 3.0)
    hn = halfnormal_dist(scale=1.5)
    assert isinstance(hn, dist.HalfNormal) and hn.scale == 1.5
    g = gamma_dist(input_vec=np.linspace(0, 10, 20))
    assert (g.concentration, g.rate) == (5.0, 1.0)
    assert (normal_dist().loc, normal_dist().scale, halfnormal_dist().scale) == (0.0, 1.0, 1.0)


def test_get_dist_errors():  # test_priors.py:141-153
    with pytest.raises(ValueError):
        uniform_dist(low=1.0)
    with pytest.raises(ValueError):
        uniform_dist(high=5.0)
    with pytest.raises(ValueError):
        uniform_dist()
    with pytest.raises(ValueError):
        gamma_dist()


@pytest.mark.parametrize("prior_type,cls", [("normal", dist.Normal), ("lognormal", dist.LogNormal)])
def test_auto_priors(prior_type, cls):  # test_priors.py:156-170
    pri = auto_priors(sample_function, 1, prior_type, loc=2.0, scale=1.0)
    assert set(pri) == {"a", "b"}
    for d in pri.values():
        assert isinstance(d, cls) and (d.loc, d.scale) == (2.0, 1.0)
    assert set(auto_normal_priors(sample_function)) == {"a", "b"}
    assert isinstance(auto_lognormal_priors(sample_function)["b"], dist.LogNormal)
    assert priors.normal_dist is normal_dist


def test_set_fn():  # test_func_setter.py:32-35
    f = set_fn(sample_function)
    assert f(2, {"a": 1, "b": 3}) == 7


def test_auto_priors_plug_into_a_model():
    from gpax_amd import ExactGP, _lib
    from gpax_amd.utils import get_keys
    from tests.oracle_engine import OracleEngine
    _lib.set_engine(OracleEngine())
    try:
        X = np.linspace(0, 3, 12)
        y = 0.5 + 2.0 * X + 0.05 * np.sin(7 * X)
        mean_fn = set_fn(lambda x, a, b: a + b * x[:, 0])
        m = ExactGP(1, "RBF", mean_fn=mean_fn, mean_fn_prior=auto_normal_priors(lambda x, a, b: None, loc=1.0, scale=2.0),
                    noise_prior_dist=halfnormal_dist(0.1), lengthscale_prior_dist=gamma_dist(2, 2))
        m.fit(get_keys()[0], X, y, num_warmup=15, num_samples=15, progress_bar=False, print_summary=False)
        s = m.get_samples()
        assert {"a", "b"} <= set(s) and s["a"].shape == (15,)
    finally:
        _lib.set_engine(None)


def _traced_site(program):
    """The one site a prior program registers when a model traces it (the analogue of numpyro.handlers.trace in
    gpax/tests/test_priors.py:42-88)."""
    from gpax_amd.infer.primitives import trace_sites
    sites, returned, _ = trace_sites(lambda: {"a": program()}, "test")
    assert len(sites) == 1 and sites[0][0] == "a" and sites[0][1] == ()
    return sites[0][2], returned["a"]


@pytest.mark.parametrize("place", ["place_normal_prior", "place_halfnormal_prior", "place_lognormal_prior"])
def test_place_prior_returns_a_value_inside_a_trace(place):  # test_priors.py:24-39
    d, value = _traced_site(lambda: getattr(priors, place)("a"))
    assert isinstance(value, np.ndarray) and np.isfinite(float(value))
    with pytest.raises(RuntimeError):
        getattr(priors, place)("a")  # there is no global tracer: outside a model's trace a site means nothing
    assert np.isfinite(float(_traced_site(lambda: priors.place_uniform_prior("a", 0, 1))[1]))
    assert np.isfinite(float(_traced_site(lambda: priors.place_gamma_prior("a", 2, 2))[1]))


def test_place_prior_params():  # test_priors.py:42-88
    d, _ = _traced_site(lambda: priors.place_normal_prior("a", loc=0.5, scale=0.1))
    assert isinstance(d, dist.Normal) and (d.loc, d.scale) == (0.5, 0.1)
    d, _ = _traced_site(lambda: priors.place_lognormal_prior("a", loc=0.5, scale=0.1))
    assert isinstance(d, dist.LogNormal) and (d.loc, d.scale) == (0.5, 0.1)
    d, _ = _traced_site(lambda: priors.place_halfnormal_prior("a", 0.1))
    assert isinstance(d, dist.HalfNormal) and d.scale == 0.1
    d, _ = _traced_site(lambda: priors.place_uniform_prior("a", low=0.5, high=1.0))
    assert isinstance(d, dist.Uniform) and (d.low, d.high) == (0.5, 1.0)
    d, _ = _traced_site(lambda: priors.place_gamma_prior("a", c=2.0, r=1.0))
    assert isinstance(d, dist.Gamma) and (d.concentration, d.rate) == (2.0, 1.0)
