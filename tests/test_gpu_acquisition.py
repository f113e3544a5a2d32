"""GPU: SURVEY.md 8f row 1 — the acquisition functions as consumers of the HIP predict / get_mvn_posterior, against
the oracle's chain: ref.predict (gp.py:351-399) -> pooled moments (acquisition.py:28-32) -> ref.acq_* (base_acq.py)."""
import numpy as np
import pytest

from gpax_amd import _lib
from gpax_amd.acquisition import EI, POI, UCB, UE
from gpax_amd.models import ExactGP, viGP
from gpax_amd.utils.utils import rng_from_key
from oracle import cpu_ref as ref
import bench_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def real_engine(engine):
    _lib.set_engine(engine)
    yield
    _lib.set_engine(None)


@pytest.mark.parametrize("kernel", ["RBF", "Matern"])
def test_mcmc_model_acquisition_on_hip_predict_matches_the_oracle_chain(kernel):
    N, d, M, S, n = 600, 2, 90, 12, 3
    X, y, Xn, _ = bench_inputs.synthetic_problem(N, d, M, seed=6)
    th = bench_inputs.synthetic_theta_samples(S, d, seed=7)
    m = ExactGP(d, kernel)
    m.X_train, m.y_train = m._set_data(X, y)
    m._samples = {k: v[None] for k, v in th.items()}  # as after fit(): (chains, S, ...)
    m._chain_shape = (1, S)
    m.mcmc = object()
    key = 42
    eps = rng_from_key(key).standard_normal((S, n, M))  # the stream predict() draws from this key
    _, y_s, _ = ref.predict(X, y, Xn, th, eps, False, kernel=kernel, jitter=1e-6, route="inv")
    mom = ref.acq_moments_from_samples(y_s, n)
    cases = [(EI, dict(maximize=True), ref.acq_ei(mom, None, True)), (EI, dict(best_f=0.2), ref.acq_ei(mom, 0.2, False)),
             (UCB, dict(beta=0.5, maximize=True), ref.acq_ucb(mom, 0.5, True)), (UCB, dict(), ref.acq_ucb(mom)),
             (POI, dict(xi=0.02), ref.acq_poi(mom, None, 0.02, False)), (UE, dict(), ref.acq_ue(mom))]
    for fn, kw, expect in cases:
        got = fn(key, m, Xn, n=n, **kw)
        assert got.shape == (M,)
        # draws agree to 1e-8; EI / POI go through (mean - best) / sigma where best is a max / min of the same vector
        np.testing.assert_allclose(got, expect, rtol=2e-6, atol=1e-9 * np.abs(expect).max())


def test_vi_model_acquisition_on_hip_posterior_matches_the_oracle():
    N, d, M = 500, 2, 70
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=8)
    m = viGP(d, "Matern")
    m.X_train, m.y_train = m._set_data(X, y)
    m.kernel_params = {"auto_loc": None}
    m.get_samples = lambda: {k: np.asarray(v) for k, v in p.items()}
    mom = ref.vigp_predict(X, y, Xn, p, False, kernel="Matern", jitter=1e-6, route="inv")
    for fn, kw, expect in [(EI, {}, ref.acq_ei(mom)), (UCB, dict(maximize=True), ref.acq_ucb(mom, 0.25, True)),
                           (POI, {}, ref.acq_poi(mom)), (UE, {}, ref.acq_ue(mom))]:
        got = fn(0, m, Xn, **kw)
        np.testing.assert_allclose(got, expect, rtol=1e-6, atol=1e-9 * np.abs(expect).max())
