"""Pins against OUTPUTS OF THE REFERENCE ITSELF.  The reference's tutorial notebook examples/gpax_simpleGP.ipynb holds,
as committed cell outputs, the NUTS posterior summaries (numpyro print_summary: mean, std, median, n_eff) that
gpax.ExactGP(1, kernel='RBF').fit(...) printed for three fully specified problems: the data come from
`np.random.seed(0)` + `np.random.uniform / normal` (25 points, noise 0.1), the model is the default one (k_length, k_scale,
noise ~ LogNormal(0, 1); jitter 1e-6), 2000 warm-up + 2000 samples.  Only those printed NUMBERS are used here (cells 14,
27 and 42 of the notebook), as expected values.

The posterior of (k_length, k_scale, noise) is three-dimensional, so it can be integrated exactly: a tensor grid in log
space, one eigendecomposition of the RBF correlation matrix per length scale (K = s R + (n + jitter) I = Q (s lam + n +
jitter) Q^T), the log likelihood tied to oracle/cpu_ref.py at random points.  What the reference printed must then agree
with the integrals of the ORACLE's model to within rounding (two decimals) plus the Monte-Carlo error its own n_eff
implies — this is what pins the restatement of the kernel (the 1/2 in the exponent, the length scaling), of the noise /
jitter placement, of the three priors and of the likelihood to the reference; a sampler is not involved.  The second
test then holds the host NUTS (gpax_amd/infer/nuts.py, on the test-only oracle engine) against the same integrals."""
import numpy as np
import pytest

from oracle import cpu_ref as ref

JITTER = 1e-6
# (mean, std, median, n_eff) as printed by the reference — examples/gpax_simpleGP.ipynb, outputs of cells 14 / 27 / 42
PRINTED = {
    "A": {"k_length": (0.17, 0.03, 0.17, 1017.15), "k_scale": (1.25, 0.90, 0.99, 850.74), "noise": (0.03, 0.02, 0.03, 1039.66)},
    "B": {"k_length": (1.16, 1.94, 0.25, 367.85), "k_scale": (0.45, 0.46, 0.30, 408.14), "noise": (0.07, 0.05, 0.07, 189.70)},
    "C": {"k_length": (0.18, 0.16, 0.15, 419.71), "k_scale": (0.32, 0.31, 0.23, 978.53), "noise": (0.04, 0.03, 0.04, 614.46)},
}


def notebook_data(case):
    """Cells 11 / 24 of the notebook: 25 points, f = sin(10 x) (A) or sin(10 x) x^2 (B, C), noise level 0.1."""
    rs = np.random.RandomState(0)  # = np.random.seed(0) followed by np.random.uniform / np.random.normal
    X = rs.uniform(-1.0, 1.0, 25)
    f = np.sin(10 * X) if case == "A" else np.sin(10 * X) * X ** 2
    return X, f + rs.normal(0.0, 0.1, 25)


def length_logprior_u(case, u):
    """Log prior density of u = log k_length up to a constant: LogNormal(0, 1), or Gamma(2, 5) in case C (cell 38:
    gpax.priors.gamma_dist(2, 5): density l exp(-5 l), times the Jacobian l)."""
    return 2.0 * u - 5.0 * np.exp(u) if case == "C" else -0.5 * u ** 2


def eig_loglik(X, y, ell, scale, noise):
    d2 = (X[:, None] - X[None, :]) ** 2
    lam, Q = np.linalg.eigh(np.exp(-0.5 * d2 / ell ** 2))
    D = scale * np.maximum(lam, 0.0) + noise + JITTER
    return float(-0.5 * ((Q.T @ y) ** 2 / D).sum() - 0.5 * np.log(D).sum() - 0.5 * X.size * np.log(2 * np.pi))


def posterior_marginals(case, nl=260, ns=200, nn=200):
    X, y = notebook_data(case)
    ul, us, un = np.linspace(-5.0, 4.0, nl), np.linspace(-7.0, 6.0, ns), np.linspace(-10.0, 3.0, nn)
    S, Nn = np.exp(us)[:, None, None], np.exp(un)[None, :, None]
    d2 = (X[:, None] - X[None, :]) ** 2
    logp = np.empty((nl, ns, nn))
    for i, u in enumerate(ul):
        lam, Q = np.linalg.eigh(np.exp(-0.5 * d2 / np.exp(u) ** 2))
        D = S * np.maximum(lam, 0.0)[None, None, :] + Nn + JITTER
        logp[i] = -0.5 * ((Q.T @ y) ** 2 / D).sum(-1) - 0.5 * np.log(D).sum(-1)
    logp += length_logprior_u(case, ul)[:, None, None] - 0.5 * (us ** 2)[None, :, None] - 0.5 * (un ** 2)[None, None, :]
    w = np.exp(logp - logp.max())
    out = {}
    for name, u, axes in (("k_length", ul, (1, 2)), ("k_scale", us, (0, 2)), ("noise", un, (0, 1))):
        w1 = w.sum(axis=axes)
        assert w1[0] + w1[-1] < 1e-5 * w1.sum()  # the box holds the posterior
        w1 = w1 / w1.sum()
        th = np.exp(u)
        mean = float((w1 * th).sum())
        out[name] = (mean, float(np.sqrt((w1 * (th - mean) ** 2).sum())), float(np.exp(np.interp(0.5, np.cumsum(w1) - 0.5 * w1, u))))  # CDF at the grid points (midpoint rule)
    return out


@pytest.fixture(scope="module")
def exact():
    return {case: posterior_marginals(case) for case in "ABC"}


def test_quadrature_likelihood_is_the_oracle_likelihood():
    rng = np.random.default_rng(0)
    for case in "AB":
        X, y = notebook_data(case)
        for _ in range(10):
            ell, s, n = np.exp(rng.normal(-1.0, 1.0)), np.exp(rng.normal(0.0, 1.0)), np.exp(rng.normal(-3.0, 1.0))
            want = ref.exactgp_log_likelihood(X[:, None], y, {"k_length": np.array([ell]), "k_scale": s, "noise": n},
                                              kernel="RBF", jitter=JITTER)
            assert abs(eig_loglik(X, y, ell, s, n) - want) <= 1e-8 * max(1.0, abs(want))


def test_oracle_posterior_reproduces_the_summaries_the_reference_printed(exact):
    for case, table in PRINTED.items():
        for name, (mean, std, median, n_eff) in table.items():
            q_mean, q_std, q_med = exact[case][name]
            half = 0.005  # two printed decimals
            se = q_std / np.sqrt(n_eff)  # Monte-Carlo error of the reference's own estimate
            assert abs(q_mean - mean) <= half + 4 * se, (case, name, "mean", q_mean, mean)
            assert abs(q_med - median) <= half + 4 * 1.2533 * se, (case, name, "median", q_med, median)
            # sample standard deviations of these heavy right tails converge slowly (and from below)
            assert -0.45 * q_std - half <= std - q_std <= 0.25 * q_std + half, (case, name, "std", q_std, std)


def test_the_pin_has_teeth():
    """The same integrals under two plausible restatement errors land far outside the tolerance: the RBF exponent
    without its 1/2 (k_length comes out a factor sqrt(2) larger), and a noise prior HalfNormal(1) instead of
    LogNormal(0, 1)."""
    X, y = notebook_data("A")
    ul, us, un = np.linspace(-5.0, 4.0, 200), np.linspace(-7.0, 6.0, 140), np.linspace(-10.0, 3.0, 140)
    S, Nn = np.exp(us)[:, None, None], np.exp(un)[None, :, None]
    d2 = (X[:, None] - X[None, :]) ** 2

    def mean_of(axis_name, half_in_exponent=True, halfnormal_noise=False):
        logp = np.empty((ul.size, us.size, un.size))
        for i, u in enumerate(ul):
            lam, Q = np.linalg.eigh(np.exp(-(0.5 if half_in_exponent else 1.0) * d2 / np.exp(u) ** 2))
            D = S * np.maximum(lam, 0.0)[None, None, :] + Nn + JITTER
            logp[i] = -0.5 * ((Q.T @ y) ** 2 / D).sum(-1) - 0.5 * np.log(D).sum(-1)
        noise_lp = (un - 0.5 * np.exp(un) ** 2) if halfnormal_noise else -0.5 * un ** 2
        logp += -0.5 * (ul ** 2)[:, None, None] - 0.5 * (us ** 2)[None, :, None] + noise_lp[None, None, :]
        w = np.exp(logp - logp.max())
        u, axes = {"k_length": (ul, (1, 2)), "noise": (un, (0, 1))}[axis_name]
        w1 = w.sum(axis=axes)
        return float((w1 / w1.sum() * np.exp(u)).sum())

    mean, _, _, n_eff = PRINTED["A"]["k_length"]
    assert abs(mean_of("k_length") - mean) < 0.01
    assert abs(mean_of("k_length", half_in_exponent=False) - mean) > 0.05
    assert abs(mean_of("noise") - PRINTED["A"]["noise"][0]) < 0.007
    assert abs(mean_of("noise", halfnormal_noise=True) - PRINTED["A"]["noise"][0]) > 0.007


def test_host_nuts_on_the_oracle_engine_matches_the_exact_posterior(exact):
    """The sampler against the integrals (no reference involved): ExactGP.fit on the notebook's first problem."""
    from gpax_amd import ExactGP, _lib
    from gpax_amd.utils import get_keys
    from tests.oracle_engine import OracleEngine

    _lib.set_engine(OracleEngine())
    try:
        X, y = notebook_data("A")
        m = ExactGP(1, kernel="RBF")
        m.fit(get_keys()[0], X, y, num_warmup=500, num_samples=1500, progress_bar=False, print_summary=False)
        s = m.get_samples()
    finally:
        _lib.set_engine(None)
    for name in ("k_length", "k_scale", "noise"):
        draws = np.asarray(s[name]).reshape(-1)
        q_mean, q_std, q_med = exact["A"][name]
        se = q_std / np.sqrt(150.0)  # a deliberately pessimistic effective sample size
        assert abs(draws.mean() - q_mean) <= 4 * se, (name, draws.mean(), q_mean)
        assert abs(np.median(draws) - q_med) <= 5 * se, (name, np.median(draws), q_med)
