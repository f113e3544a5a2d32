"""
oracle/cpu_ref.py — CPU restatement (NumPy/SciPy, fp64) of gpax's exact-GP hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under gpax_amd/ imports this file; only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may, and only as the checker /
the reported CPU baseline — never as the thing measured or shipped.

PARITY: PINNED TO REFERENCE-HELD OUTPUTS AT THE LEVEL OF POSTERIOR SUMMARIES AND OF THE LOG-JOINT VALUE, NOT OF VECTORS.
The reference (ziatdinovmax/gpax v0.1.9) is pure Python on JAX + NumPyro; neither jax, jaxlib nor numpyro is installed in
the build container (and they never travel to the GPU box), so the reference cannot be imported to generate vectors, and
its own tests pin no numeric value on this path.  What the reference does hold are the committed cell
outputs of its tutorial notebooks (examples/gpax_simpleGP.ipynb, gpax_UIGP.ipynb, MeasuredNoiseGP.ipynb, gpax_GPBO.ipynb,
GP_sGP.ipynb, simpleGP.ipynb, compare_GPs.ipynb): the NUTS posterior summaries (mean, std, median, n_eff of k_length, k_scale, noise) gpax printed for
ten problems whose data are fixed by np.random.seed(k) — ExactGP with RBF, Matern and Periodic kernels, LogNormal /
Gamma / HalfNormal priors, MeasuredNoiseGP — and the point estimate and loss of a viGP fit after 1000 SVI steps.
tests/test_reference_notebook_pins.py integrates the two- / three-dimensional posteriors of THIS file's model exactly
(tensor grid; correlation matrices from the kernel functions below; likelihood tied to exactgp_log_likelihood) and
requires all 87 printed numbers to agree within their two decimals plus the Monte-Carlo error the printed n_eff implies,
requires the negative log joint at the printed viGP state to sit within 0.04 of the printed loss, and shows that plausible
restatement errors (no 1/2 in the RBF exponent, RBF or Matern-3/2 for Matern-5/2, a Matern-5/2 without its quadratic term,
another noise prior, measured variances left out) fail.  That pins kernel formulas, noise / jitter / measured-noise placement, priors and
the likelihood (its constants included) to the reference statistically — k_length to 1 - 5 %, the log joint to 0.3 %.
The PREDICTIVE leg (get_mvn_posterior's mean and marginal variance, pooled over posterior samples, and the acquisition arithmetic on top)
is pinned the same way through the seven posterior tables examples/gpax_GPBO.ipynb cell 22 printed for its Bayesian-optimisation
loop — each step's data contain the argmax of UCB over predict() of the step before: tests/test_reference_gpbo_loop.py runs that loop
exactly with this file's kernel functions and reproduces the tables (steps 4 - 7 within the path envelope the reference's own Monte-Carlo
error leaves).  Covariances off the diagonal, MVN sampling given eps, the sparse (VFE) bound / posterior and everything bit-level remain
UNPINNED by reference-generated numbers.  The sparse leg is tied to the pinned exact leg ANALYTICALLY only
(tests/test_oracle.py::test_sparse_bound_tends_to_the_exact_log_likelihood_when_inducing_equals_train and
::test_sparse_posterior_tends_to_exact_when_inducing_equals_train): at Xu = X the reference's own calls make sparse_bound equal
exactgp_log_likelihood (the pinned function) to 1e-7, and nested inducing sets approach it from below.  That pins the
LowRankMVN log-density, the Woodbury algebra and the SIGN and clip of the trace term at that limit; it does NOT pin the
trace term's magnitude away from the limit (it vanishes there), nor any number the reference generated for a sparse model.  Beyond that, this restatement follows the
reference line by line (file:line cited per function, paths relative to the reference checkout) and is cross-checked
independently (tests/test_oracle.py): direct-formula Gram in mpmath at 50 digits, scipy.stats.multivariate_normal for
the log-density, explicit-inverse vs Cholesky route for the posterior, central finite differences for the gradient,
dense W W^T + D formulation for the low-rank MVN.  The third-party arithmetic reached by the path (NumPyro
MultivariateNormal / LowRankMultivariateNormal log_prob and sample; jnp.linalg.inv / cholesky; pins jax>=0.6.2,
numpyro>=0.18.0 in the reference's pyproject.toml:26-28) is restated from its published definitions and marked [knowledge].
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import scipy.linalg as sla

LOG_2PI = math.log(2.0 * math.pi)


# ------------------------------------------------------------------------------------------
# gpax/kernels/kernels.py
# ------------------------------------------------------------------------------------------
def _sqrt(x, eps=1e-12):
    """gpax/kernels/kernels.py:20-21"""
    return np.sqrt(x + eps)


def add_jitter(x, jitter=1e-6):
    """gpax/kernels/kernels.py:24-25"""
    return x + jitter


def square_scaled_distance(X, Z, lengthscale=1.0):
    """gpax/kernels/kernels.py:28-41 — expansion form with the clip at 0, op for op."""
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    scaled_X = X / lengthscale
    scaled_Z = Z / lengthscale
    X2 = (scaled_X ** 2).sum(1, keepdims=True)
    Z2 = (scaled_Z ** 2).sum(1, keepdims=True)
    XZ = np.matmul(scaled_X, scaled_Z.T)
    r2 = X2 - 2 * XZ + Z2.T
    return r2.clip(0)


def RBFKernel(X, Z, params, noise=0, jitter=1e-6, **kwargs):
    """gpax/kernels/kernels.py:44-65"""
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    r2 = square_scaled_distance(X, Z, params["k_length"])
    k = params["k_scale"] * np.exp(-0.5 * r2)
    if X.shape == Z.shape:
        k = k + add_jitter(noise, jitter) * np.eye(X.shape[0])
    return k


def MaternKernel(X, Z, params, noise=0, jitter=1e-6, **kwargs):
    """gpax/kernels/kernels.py:68-91 (Matern-5/2)"""
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    r2 = square_scaled_distance(X, Z, params["k_length"])
    r = _sqrt(r2)
    sqrt5_r = 5 ** 0.5 * r
    k = params["k_scale"] * (1 + sqrt5_r + (5 / 3) * r2) * np.exp(-sqrt5_r)
    if X.shape == Z.shape:
        k = k + add_jitter(noise, jitter) * np.eye(X.shape[0])
    return k


def PeriodicKernel(X, Z, params, noise=0, jitter=1e-6, **kwargs):
    """gpax/kernels/kernels.py:94-117"""
    X = np.asarray(X, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    d = X[:, None] - Z[None]
    scaled_sin = np.sin(math.pi * d / params["period"]) / params["k_length"]
    k = params["k_scale"] * np.exp(-2 * (scaled_sin ** 2).sum(-1))
    if X.shape == Z.shape:
        k = k + add_jitter(noise, jitter) * np.eye(X.shape[0])
    return k


def get_kernel(kernel="RBF"):
    """gpax/kernels/kernels.py:227-241 (in-scope names only)"""
    kernel_book = {"RBF": RBFKernel, "Matern": MaternKernel, "Periodic": PeriodicKernel}
    if isinstance(kernel, str):
        return kernel_book[kernel]
    return kernel


def noise_kernel(name):
    """gpax/utils/fn.py:119-147 `_set_noise_kernel_fn`: the named kernel reading k_noise_length / k_noise_scale."""
    base = get_kernel(name)

    def k(X, Z, params, noise=0, jitter=1e-6, **kwargs):
        p = dict(params)
        p["k_length"], p["k_scale"] = params["k_noise_length"], params["k_noise_scale"]
        return base(X, Z, p, noise, jitter=jitter, **kwargs)
    return k


def varnoise_log_likelihood(X, y, params, kernel="RBF", noise_kernel_name="RBF", jitter=1e-6, noise_loc=None,
                            f_loc=None) -> float:
    """The two MVN terms of VarNoiseGP.model, gpax/models/hskgp.py:129-153:
    log MVN(log_var; noise_loc, k_noise) + log MVN(y; f_loc, k + diag(exp(log_var)))."""
    X, y = _set_data(X, y)
    N = X.shape[0]
    lv = np.asarray(params["log_var"], dtype=np.float64)
    k_noise = noise_kernel(noise_kernel_name)(X, X, params, 0, jitter=jitter)
    t1 = mvn_log_prob(lv, np.zeros(N) if noise_loc is None else noise_loc, k_noise)
    k = get_kernel(kernel)(X, X, params, 0, jitter=jitter)
    t2 = mvn_log_prob(y, np.zeros(N) if f_loc is None else f_loc, k + np.diag(np.exp(lv)))
    return t1 + t2


def varnoise_get_mvn_posterior(X_train, y_train, X_new, params, kernel="RBF", noise_kernel_name="RBF", jitter=1e-6,
                               noise_loc_train=None, noise_loc_new=None):
    """VarNoiseGP.get_mvn_posterior, gpax/models/hskgp.py:167-206 (explicit inverses, as there)."""
    X_train, y_train = _set_data(X_train, y_train)
    X_new = _set_data(X_new)
    kfn, nfn = get_kernel(kernel), noise_kernel(noise_kernel_name)
    k_pp = kfn(X_new, X_new, params, 0, jitter=jitter)
    k_pX = kfn(X_new, X_train, params, jitter=0.0)
    k_XX = kfn(X_train, X_train, params, 0, jitter=jitter)
    K_xx_inv = np.linalg.inv(k_XX)
    cov = k_pp - k_pX @ (K_xx_inv @ k_pX.T)
    mean = k_pX @ (K_xx_inv @ y_train)
    k_pX_noise = nfn(X_new, X_train, params, jitter=0.0)
    k_XX_noise = nfn(X_train, X_train, params, 0, jitter=jitter)
    lv_res = np.asarray(params["log_var"], dtype=np.float64).copy()
    if noise_loc_train is not None:
        lv_res = lv_res - noise_loc_train
    plv = k_pX_noise @ (np.linalg.inv(k_XX_noise) @ lv_res)
    if noise_loc_new is not None:
        plv = plv + noise_loc_new
    return mean, cov + np.diag(np.exp(plv))


def measured_noise_predict_one(X_train, y_train, X_new, params, noise_predicted, eps, noiseless=True, kernel="RBF",
                               jitter=1e-6):
    """MeasuredNoiseGP._predict, gpax/models/mngp.py:159-181, given the standard normals eps (n, M):
    (mean, cov) = ExactGP.get_mvn_posterior with the deterministic noise = 0 (the training block carries NO
    measured noise there), cov += diag(noise_predicted), draws from the marginals only."""
    p = dict(params)
    p["noise"] = 0.0
    mean, cov = get_mvn_posterior(X_train, y_train, X_new, p, noiseless, kernel=kernel, jitter=jitter, route="inv")
    sig = np.sqrt(np.clip(np.diag(cov) + np.asarray(noise_predicted, dtype=np.float64), 0.0, None))
    return mean, mean[None, :] + sig[None, :] * np.asarray(eps, dtype=np.float64)


# ------------------------------------------------------------------------------------------
# NumPyro MultivariateNormal [knowledge]: log_prob = -1/2 |L^-1 (y-loc)|^2 - sum log L_ii
#   - n/2 log 2 pi with L = cholesky(covariance_matrix); sample = loc + L @ eps.
#   Source of the formula: numpyro/distributions/continuous.py, class MultivariateNormal — __init__ sets
#   scale_tril = cholesky(covariance_matrix); log_prob computes M = _batch_mahalanobis(scale_tril, value - loc)
#   (a solve_triangular followed by a sum of squares), half_log_det = sum(log(diagonal(scale_tril))),
#   normalize_term = half_log_det + 0.5 * event_size * log(2 pi), result -0.5 * M - normalize_term; sample returns
#   loc + squeeze(scale_tril @ eps[..., None]) with eps ~ N(0, 1) of shape sample_shape + batch_shape + event_shape.
#   Pinned here (tests/test_oracle.py) against scipy.stats.multivariate_normal.logpdf, an independent
#   implementation of the same textbook density (Rasmussen & Williams, GPML eq. 2.30 / A.9).
# ------------------------------------------------------------------------------------------
def mvn_log_prob(y, loc, cov) -> float:
    y = np.asarray(y, dtype=np.float64)
    try:
        L = np.linalg.cholesky(cov)
    except np.linalg.LinAlgError:
        return float("nan")
    w = sla.solve_triangular(L, y - loc, lower=True)
    return float(-0.5 * (w @ w) - np.log(np.diag(L)).sum() - 0.5 * y.shape[0] * LOG_2PI)


def mvn_sample(loc, cov, eps):
    """dist.MultivariateNormal(loc, cov).sample given the standard normals eps (n, M)."""
    try:
        L = np.linalg.cholesky(cov)
    except np.linalg.LinAlgError:
        return np.full((np.asarray(eps).shape[0], loc.shape[0]), np.nan)
    return loc[None, :] + np.asarray(eps) @ L.T


# ------------------------------------------------------------------------------------------
# gpax/models/gp.py
# ------------------------------------------------------------------------------------------
def _set_data(X, y=None):
    """gpax/models/gp.py:410-414"""
    X = np.asarray(X, dtype=np.float64)
    X = X if X.ndim > 1 else X[:, None]
    if y is not None:
        return X, np.asarray(y, dtype=np.float64).squeeze()
    return X


def exactgp_log_likelihood(X, y, params, kernel="RBF", jitter=1e-6, mean_fn=None, mean_params=None,
                           measured_noise=None) -> float:
    """log p(y | theta) of ExactGP.model, gpax/models/gp.py:137-164 (the `y` site only).
    measured_noise: MeasuredNoiseGP.model, gpax/models/mngp.py:92-98 — covariance k + diag(measured_noise)
    (called there with params['noise'] = 0)."""
    X, y = _set_data(X, y)
    kfn = get_kernel(kernel)
    f_loc = np.zeros(X.shape[0])
    if mean_fn is not None:
        args = [X] if mean_params is None else [X, mean_params]
        f_loc = f_loc + np.asarray(mean_fn(*args)).squeeze()
    k = kfn(X, X, params, params["noise"], jitter=jitter)
    if measured_noise is not None:
        k = k + np.diag(np.asarray(measured_noise, dtype=np.float64))
    return mvn_log_prob(y, f_loc, k)


def exactgp_log_likelihood_grad(X, y, params, kernel="RBF", jitter=1e-6, yres=None, measured_noise=None):
    """Analytic gradient of the log-likelihood w.r.t. (k_length[d], k_scale, noise) and
    alpha = K^-1 yres: 1/2 sum_ij (alpha alpha^T - K^-1)_ij dK_ij/dtheta.  (The reference gets
    this from JAX autodiff through gp.py:137-164; tests check it against central differences.)"""
    X, y = _set_data(X, y)
    if yres is None:
        yres = y
    N, d = X.shape
    ell = np.broadcast_to(np.asarray(params["k_length"], dtype=np.float64).reshape(-1), (d,)).astype(np.float64)
    s = float(params["k_scale"])
    kfn = get_kernel(kernel)
    K = kfn(X, X, params, params["noise"], jitter=jitter)
    if measured_noise is not None:
        K = K + np.diag(np.asarray(measured_noise, dtype=np.float64))
    Kinv = np.linalg.inv(K)
    alpha = Kinv @ yres
    G = np.outer(alpha, alpha) - Kinv
    diff = (X[:, None, :] - X[None, :, :]) / ell  # u_m
    r2 = (diff ** 2).sum(-1)
    if kernel == "RBF":
        kb = s * np.exp(-0.5 * r2)
        dk = -0.5 * kb
    else:
        r = np.sqrt(r2 + 1e-12)
        e = np.exp(-math.sqrt(5.0) * r)
        kb = s * (1 + math.sqrt(5.0) * r + (5 / 3) * r2) * e
        dk = -(5.0 / 6.0) * s * e * (1 + math.sqrt(5.0) * r2 / r)
    g_ell = np.array([0.5 * np.sum(G * dk * (-2.0 * diff[:, :, m] ** 2 / ell[m])) for m in range(d)])
    g_scale = 0.5 * np.sum(G * kb) / s
    g_noise = 0.5 * np.trace(G)
    return g_ell, g_scale, g_noise, alpha


def exactgp_log_likelihood_grad_blocked(X, y, params, kernel="RBF", jitter=1e-6, block=1024):
    """exactgp_log_likelihood_grad for sizes where its (N, N, d) temporaries do not fit (C3: N = 16384 would need
    ~15 GB): the same formula, 1/2 sum_ij (alpha alpha^T - K^-1)_ij dK_ij/dtheta, with K^-1 from the Cholesky factor
    (LAPACK potri) and the contraction accumulated over row blocks.  Peak memory 2 N^2 doubles.  tests/test_oracle.py
    holds it against the unblocked function; the GPU test of the headline size uses it."""
    from scipy.linalg import lapack
    X, y = _set_data(X, y)
    N, d = X.shape
    ell = np.broadcast_to(np.asarray(params["k_length"], dtype=np.float64).reshape(-1), (d,)).astype(np.float64)
    s = float(params["k_scale"])
    K = get_kernel(kernel)(X, X, params, params["noise"], jitter=jitter)
    c, info = lapack.dpotrf(K, lower=1, overwrite_a=1)
    assert info == 0
    alpha = sla.cho_solve((c, True), y)
    Kinv, info = lapack.dpotri(c, lower=1, overwrite_c=1)  # lower triangle of K^-1
    assert info == 0
    g_ell = np.zeros(d)
    g_scale = 0.0
    g_trace = 0.0
    for r0 in range(0, N, block):
        r1 = min(N, r0 + block)
        # row block of G = alpha alpha^T - K^-1 (symmetric: the block's upper part comes from the transposed column block)
        Kb = np.tril(Kinv[r0:r1, :], k=r0)  # entries (i, j) with j <= i
        Kb[:, r0:] += np.triu(Kinv[r0:, r0:r1].T, k=1)[:, : N - r0]
        G = np.outer(alpha[r0:r1], alpha) - Kb
        g_trace += float(np.trace(G[:, r0:r1]))
        diff = (X[r0:r1, None, :] - X[None, :, :]) / ell
        r2 = (diff ** 2).sum(-1)
        if kernel == "RBF":
            kb = s * np.exp(-0.5 * r2)
            dk = -0.5 * kb
        else:
            r = np.sqrt(r2 + 1e-12)
            e = np.exp(-math.sqrt(5.0) * r)
            kb = s * (1 + math.sqrt(5.0) * r + (5 / 3) * r2) * e
            dk = -(5.0 / 6.0) * s * e * (1 + math.sqrt(5.0) * r2 / r)
        Gd = G * dk
        for m in range(d):
            g_ell[m] += 0.5 * float(np.sum(Gd * (-2.0 * diff[:, :, m] ** 2 / ell[m])))
        g_scale += 0.5 * float(np.sum(G * kb)) / s
    return g_ell, g_scale, 0.5 * g_trace, alpha


def get_mvn_posterior(X_train, y_train, X_new, params, noiseless=False, kernel="RBF", jitter=1e-6,
                      mean_fn=None, mean_fn_has_params=False, route="inv") -> Tuple[np.ndarray, np.ndarray]:
    """ExactGP.get_mvn_posterior, gpax/models/gp.py:253-277.
    route='inv' is the reference's explicit inverse (gp.py:271-273); route='chol' is the
    POTRF+TRSM route the GPU runs.  tests assert the two agree."""
    X_train, y_train = _set_data(X_train, y_train)
    X_new = _set_data(X_new)
    kfn = get_kernel(kernel)
    noise = params["noise"]
    noise_p = noise * (1 - int(bool(noiseless)))
    y_residual = y_train.copy()
    if mean_fn is not None:
        args = [X_train, params] if mean_fn_has_params else [X_train]
        y_residual = y_residual - np.asarray(mean_fn(*args)).squeeze()
    k_pp = kfn(X_new, X_new, params, noise_p, jitter=jitter)
    k_pX = kfn(X_new, X_train, params, jitter=0.0)
    k_XX = kfn(X_train, X_train, params, noise, jitter=jitter)
    if route == "inv":
        K_xx_inv = np.linalg.inv(k_XX)
        cov = k_pp - np.matmul(k_pX, np.matmul(K_xx_inv, np.transpose(k_pX)))
        mean = np.matmul(k_pX, np.matmul(K_xx_inv, y_residual))
    else:
        L = np.linalg.cholesky(k_XX)
        V = sla.solve_triangular(L, k_pX.T, lower=True)
        w = sla.solve_triangular(L, y_residual, lower=True)
        cov = k_pp - V.T @ V
        mean = V.T @ w
    if mean_fn is not None:
        args = [X_new, params] if mean_fn_has_params else [X_new]
        mean = mean + np.asarray(mean_fn(*args)).squeeze()
    return mean, cov


def predict_one(X_train, y_train, X_new, params, eps, noiseless=False, kernel="RBF", jitter=1e-6, **kw):
    """ExactGP._predict, gpax/models/gp.py:279-293, with the MVN draw made explicit in eps (n, M)."""
    y_mean, K = get_mvn_posterior(X_train, y_train, X_new, params, noiseless, kernel, jitter, **kw)
    return y_mean, mvn_sample(y_mean, K, eps)


def exactgp_full_pass(X_train, y_train, X_new, params, eps, noiseless=False, kernel="RBF", jitter=1e-6):
    """lml (gp.py:137-164), posterior mean / cov (gp.py:253-277, Cholesky route) and the MVN draw (gp.py:279-293)
    of ONE theta from ONE factorisation of k_XX — the same arithmetic as exactgp_log_likelihood +
    get_mvn_posterior(route='chol') + mvn_sample (tests/test_oracle.py asserts equality), for the full-size
    parity tests where three separate N = 16384 factorisations on the host would take minutes."""
    X_train, y_train = _set_data(X_train, y_train)
    X_new = _set_data(X_new)
    kfn = get_kernel(kernel)
    noise = params["noise"]
    noise_p = noise * (1 - int(bool(noiseless)))
    k_XX = kfn(X_train, X_train, params, noise, jitter=jitter)
    L = np.linalg.cholesky(k_XX)
    del k_XX
    w = sla.solve_triangular(L, y_train, lower=True)
    lml = float(-0.5 * (w @ w) - np.log(np.diag(L)).sum() - 0.5 * y_train.shape[0] * LOG_2PI)
    k_pX = kfn(X_new, X_train, params, jitter=0.0)
    V = sla.solve_triangular(L, k_pX.T, lower=True)
    k_pp = kfn(X_new, X_new, params, noise_p, jitter=jitter)
    cov = k_pp - V.T @ V
    mean = V.T @ w
    alpha = sla.solve_triangular(L, w, lower=True, trans="T")
    return lml, mean, cov, mvn_sample(mean, cov, eps), alpha


def predict(X_train, y_train, X_new, samples: Dict[str, np.ndarray], eps, noiseless=False, kernel="RBF",
            jitter=1e-6, **kw):
    """ExactGP.predict, gpax/models/gp.py:351-399: the vmap over S samples as a loop.
    eps has shape (S, n, M).  Returns (y_means.mean(0), y_sampled (S, n, M), y_means (S, M))."""
    S = len(next(iter(samples.values())))
    means, draws = [], []
    for s in range(S):
        prm = {k: np.asarray(v)[s] for k, v in samples.items()}
        m, y = predict_one(X_train, y_train, X_new, prm, eps[s], noiseless, kernel, jitter, **kw)
        means.append(m)
        draws.append(y)
    means = np.stack(means)
    return means.mean(0), np.stack(draws), means


def vigp_predict(X_train, y_train, X_new, params, noiseless=False, kernel="RBF", jitter=1e-6, **kw):
    """viGP.predict, gpax/models/vigp.py:153-185: (mean, cov.diagonal())."""
    mean, cov = get_mvn_posterior(X_train, y_train, X_new, params, noiseless, kernel, jitter, **kw)
    return mean, cov.diagonal()


# ------------------------------------------------------------------------------------------
# gpax/models/sparse_gp.py  +  NumPyro LowRankMultivariateNormal [knowledge]
# ------------------------------------------------------------------------------------------
def lowrank_mvn_log_prob(y, loc, W, D) -> float:
    """LowRankMultivariateNormal(loc, cov_factor=W (N,M), cov_diag=D (N,)).log_prob(y):
    Woodbury + matrix-determinant lemma with the capacitance matrix C = I + W^T D^-1 W.
    Source of the formula [knowledge]: numpyro/distributions/continuous.py, class LowRankMultivariateNormal —
    __init__: Wt_Dinv = W^T / D, K = I + Wt_Dinv W, _capacitance_tril = cholesky(K); log_prob:
    M = _batch_lowrank_mahalanobis(W, D, diff, capacitance_tril) = sum(diff^2 / D) - |C_tril^-1 (Wt_Dinv diff)|^2,
    log_det = _batch_lowrank_logdet(...) = 2 sum log diag(C_tril) + sum log D, result -0.5 (n log 2 pi + log_det + M)
    (the same decomposition as torch.distributions.LowRankMultivariateNormal, from which NumPyro's was ported).
    Pinned (tests/test_oracle.py) against the dense MVN density of W W^T + diag(D) to 1e-11."""
    y = np.asarray(y, dtype=np.float64)
    r = y - loc
    Wt_Dinv = W.T / D
    Cm = np.eye(W.shape[1]) + Wt_Dinv @ W
    Lc = np.linalg.cholesky(Cm)
    Wt_Dinv_r = Wt_Dinv @ r
    t = sla.solve_triangular(Lc, Wt_Dinv_r, lower=True)
    maha = (r * r / D).sum() - t @ t
    logdet = np.log(D).sum() + 2.0 * np.log(np.diag(Lc)).sum()
    return float(-0.5 * (y.shape[0] * LOG_2PI + logdet + maha))


def sparse_bound(X, y, Xu, params, kernel="Matern", jitter=1e-6, f_loc=None) -> float:
    """viSparseGP.model, gpax/models/sparse_gp.py:62-114: VFE bound =
    LowRankMVN log_prob - trace_term / 2 (the `y` site plus the `trace_term` factor)."""
    X, y = _set_data(X, y)
    kfn = get_kernel(kernel)
    noise = params["noise"]
    N = X.shape[0]
    D = np.broadcast_to(noise, (N,)).astype(np.float64)
    loc = np.zeros(N) if f_loc is None else f_loc
    Kuu = kfn(Xu, Xu, params, jitter=jitter)
    Luu = np.linalg.cholesky(Kuu)  # cholesky(Kuu).T of the upper factor == lower factor
    Kuf = kfn(Xu, X, params)
    W = sla.solve_triangular(Luu, Kuf, lower=True).T
    Kffdiag = np.diag(kfn(X, X, params, jitter=0))
    Qffdiag = np.square(W).sum(axis=-1)
    trace_term = (Kffdiag - Qffdiag).sum() / noise
    trace_term = np.clip(trace_term, 0, None)
    return lowrank_mvn_log_prob(y, loc, W, D) - trace_term / 2.0


def sparse_posterior(X_train, y_train, Xu, X_new, params, noiseless=False, kernel="Matern", jitter=1e-6, mean_fn=None,
                     mean_fn_has_params=False):
    """viSparseGP.get_mvn_posterior, gpax/models/sparse_gp.py:173-223; mean_fn / mean_fn_has_params as
    sparse_gp.py:189-192 (residual) and :219-221 (mean added back at X_new)."""
    X_train, y_train = _set_data(X_train, y_train)
    X_new = _set_data(X_new)
    kfn = get_kernel(kernel)
    noise = params["noise"]
    N = X_train.shape[0]
    D = np.broadcast_to(noise, (N,)).astype(np.float64)
    noise_p = noise * (1 - int(bool(noiseless)))
    y_residual = y_train.copy()
    if mean_fn is not None:
        args = [X_train, params] if mean_fn_has_params else [X_train]
        y_residual = y_residual - np.asarray(mean_fn(*args)).squeeze()
    Kuu = kfn(Xu, Xu, params, jitter=jitter)
    Luu = np.linalg.cholesky(Kuu)
    Kuf = kfn(Xu, X_train, params, jitter=0)
    W = sla.solve_triangular(Luu, Kuf, lower=True)
    W_Dinv = W / D
    K = W_Dinv @ W.T
    K[np.diag_indices(K.shape[0])] += 1
    L = np.linalg.cholesky(K)
    y_2D = y_residual.reshape(-1, N).T
    W_Dinv_y = W_Dinv @ y_2D
    Kus = kfn(Xu, X_new, params, jitter=0)
    Ws = sla.solve_triangular(Luu, Kus, lower=True)
    pack = np.concatenate((W_Dinv_y, Ws), axis=1)
    Linv_pack = sla.solve_triangular(L, pack, lower=True)
    Linv_W_Dinv_y = Linv_pack[:, : W_Dinv_y.shape[1]]
    Linv_Ws = Linv_pack[:, W_Dinv_y.shape[1]:]
    mean = (Linv_W_Dinv_y.T @ Linv_Ws).squeeze()
    Kss = kfn(X_new, X_new, params, noise_p, jitter=jitter)
    Qss = Ws.T @ Ws
    cov = Kss - Qss + Linv_Ws.T @ Linv_Ws
    if mean_fn is not None:
        args = [X_new, params] if mean_fn_has_params else [X_new]
        mean = mean + np.asarray(mean_fn(*args)).squeeze()
    return mean, cov


# ------------------------------------------------------------------------------------------
# gpax/utils/utils.py (host plumbing the build's API reproduces)
# ------------------------------------------------------------------------------------------
def split_in_batches(X_new, batch_size=100, dim=0):
    """gpax/utils/utils.py:33-51 (including its UnboundLocalError when len < batch_size)."""
    if dim not in [0, 1]:
        raise NotImplementedError("'dim' must be equal to 0 or 1")
    num_batches = X_new.shape[dim] // batch_size
    X_split = []
    for i in range(num_batches):
        X_i = X_new[i * batch_size:(i + 1) * batch_size] if dim == 0 else X_new[:, i * batch_size:(i + 1) * batch_size]
        X_split.append(X_i)
    X_i = X_new[(i + 1) * batch_size:] if dim == 0 else X_new[:, (i + 1) * batch_size:]
    if X_i.shape[dim] > 0:
        X_split.append(X_i)
    return X_split


def split_dict(data, chunk_size):
    """gpax/utils/utils.py:54-81"""
    N = len(next(iter(data.values())))
    num_chunks = int(np.ceil(N / chunk_size))
    result = []
    for i in range(num_chunks):
        s, e = i * chunk_size, min((i + 1) * chunk_size, N)
        result.append({k: v[s:e] for k, v in data.items()})
    return result


def preprocess_sparse_image(sparse_image):
    """gpax/utils/utils.py:150-168"""
    dtype = sparse_image.dtype
    non_zero_indices = np.nonzero(sparse_image)
    gp_input = np.column_stack(non_zero_indices)
    targets = sparse_image[non_zero_indices]
    full_indices = np.array(np.meshgrid(*[np.arange(dim) for dim in sparse_image.shape])).T.reshape(
        -1, sparse_image.ndim)
    return gp_input.astype(dtype), targets.astype(dtype), full_indices.astype(dtype)


# ------------------------------------------------------------------------------------------
# gpax/acquisition/base_acq.py:20-155 and acquisition.py:22-35
# NumPyro Normal(0, 1) [knowledge]: cdf(u) = (1 + erf(u / sqrt 2)) / 2, log_prob(u) = -u^2 / 2 - log sqrt(2 pi)
# ------------------------------------------------------------------------------------------
def _std_normal_cdf(u):
    from scipy.special import erf
    return 0.5 * (1.0 + erf(u / math.sqrt(2.0)))


def _std_normal_log_prob(u):
    return -0.5 * u * u - 0.5 * LOG_2PI


def acq_ei(moments, best_f=None, maximize=False):
    """base_acq.py:20-73"""
    mean, var = moments
    if best_f is None:
        best_f = mean.max() if maximize else mean.min()
    sigma = np.sqrt(var)
    u = (mean - best_f) / sigma
    if not maximize:
        u = -u
    ucdf = _std_normal_cdf(u)
    updf = np.exp(_std_normal_log_prob(u))
    return sigma * (updf + u * ucdf)


def acq_ucb(moments, beta=0.25, maximize=False):
    """base_acq.py:76-109"""
    mean, var = moments
    delta = np.sqrt(beta * var)
    if maximize:
        return mean + delta
    return -(mean - delta)


def acq_ue(moments):
    """base_acq.py:112-133 (returns the standard deviation)"""
    _, var = moments
    return np.sqrt(var)


def acq_poi(moments, best_f=None, xi=0.01, maximize=False):
    """base_acq.py:136-155"""
    mean, var = moments
    if best_f is None:
        best_f = mean.max() if maximize else mean.min()
    sigma = np.sqrt(var)
    u = (mean - best_f - xi) / sigma
    if not maximize:
        u = -u
    return _std_normal_cdf(u)


def acq_moments_from_samples(y_sampled, n):
    """acquisition.py:28-32: pooled moments of the (S, n, M) predictive draws of an MCMC model."""
    y_sampled = y_sampled.reshape(n * y_sampled.shape[0], -1)
    return y_sampled.mean(0), y_sampled.var(0)
