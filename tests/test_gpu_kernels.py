"""GPU parity tests of the individual HIP kernels / drivers, through the C-ABI (ctypes).

Checker: plain NumPy/SciPy fp64 and oracle/cpu_ref.py on the same seeded inputs.
Tolerances are fp64 round-off scaled by the problem's conditioning and are written per test.
"""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


# ---- MFMA GEMM tile kernel ---------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(16, 16, 4), (128, 128, 16), (200, 136, 72), (256, 384, 512), (130, 129, 17)])
def test_gemm_nt_matches_numpy(engine, M, N, K):
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    # asymmetric operands: a transposed C-write or swapped fragment layout cannot pass
    A = rng.standard_normal((M, K)) + np.arange(M)[:, None] * 0.01
    B = rng.standard_normal((N, K)) - np.arange(N)[:, None] * 0.02
    Cgpu = engine.gemm_nt(A, B)
    assert rel(Cgpu, A @ B.T) < 1e-14


def test_gemm_nt_identity_asymmetric(engine):
    # A = I picks out rows of B^T exactly: checks the f64 MFMA lane map bit for bit
    n = 128
    B = np.arange(n * n, dtype=np.float64).reshape(n, n) * 0.5 + 1.0
    C = engine.gemm_nt(np.eye(n), B)
    np.testing.assert_array_equal(C, B.T)


def test_gemm_nt_alpha_beta(engine):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((150, 40))
    B = rng.standard_normal((70, 40))
    C0 = rng.standard_normal((150, 70))
    C = engine.gemm_nt(A, B, alpha=-1.0, beta=1.0, Cin=C0)
    assert rel(C, C0 - A @ B.T) < 1e-14


# ---- blocked Cholesky --------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 7, 100, 128, 129, 300, 640, 1000])
def test_potrf_matches_lapack(engine, n):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 3))
    A = G @ G.T + n * 1e-3 * np.eye(n)
    L, info = engine.potrf(A)
    assert info == 0
    Lref = np.linalg.cholesky(A)
    assert rel(L, Lref) < 1e-12
    assert rel(L @ L.T, A) < 1e-14
    assert np.all(np.triu(L, 1) == 0)


def test_potrf_not_positive_definite_reports_info(engine):
    rng = np.random.default_rng(0)
    G = rng.standard_normal((200, 200))
    A = G @ G.T
    A[150, 150] = -1.0  # pivot 151 fails (or earlier through fill-in)
    L, info = engine.potrf(A)
    assert 0 < info <= 151


# ---- Gram kernels ------------------------------------------------------------------------
@pytest.mark.parametrize("kind,name", [(0, "RBF"), (1, "Matern")])
@pytest.mark.parametrize("n,m,d", [(5, 5, 1), (64, 37, 2), (257, 513, 3), (33, 1030, 4), (40, 50, 6)])
def test_gram_matches_oracle(engine, kind, name, n, m, d):
    rng = np.random.default_rng(n + m + d)
    X = rng.uniform(0, 10, (n, d))
    Z = rng.uniform(0, 10, (m, d)) if (n, m) != (5, 5) else X.copy()
    ell = 0.5 + rng.uniform(0, 2, d)
    params = {"k_length": ell, "k_scale": 1.7}
    add_diag = X.shape == Z.shape
    K = engine.gram(kind, X, Z, ell, 1.7, 0.1 + 1e-6, add_diag)
    Kref = ref.get_kernel(name)(X, Z, params, noise=0.1, jitter=1e-6)
    # the oracle uses the reference's expansion formula (cancellation ~1e-16 |x/l|^2), the GPU
    # the direct difference: agreement is bounded by that cancellation, not by 1 ulp
    np.testing.assert_allclose(K, Kref, rtol=1e-10, atol=1e-12)


def test_gram_coincident_points_and_scalar_lengthscale(engine):
    X = np.array([[0.0, 1.0], [0.0, 1.0], [2.0, 3.0]])
    for kind, name in [(0, "RBF"), (1, "Matern")]:
        K = engine.gram(kind, X, X, 1.3, 2.0, 0.5, True)
        Kref = ref.get_kernel(name)(X, X, {"k_length": 1.3, "k_scale": 2.0}, noise=0.5 - 1e-6, jitter=1e-6)
        np.testing.assert_allclose(K, Kref, rtol=1e-12, atol=1e-12)
