"""gpax_amd.utils.threefry: the block function against the Random123 / JAX known-answer vectors, and the layers above
it (PRNGKey / split / normal) against the values JAX's documentation publishes, in both counter layouts
(jax_threefry_partitionable True = default since JAX 0.5, and the legacy one)."""
import numpy as np
import pytest

from gpax_amd.utils import threefry as tf


@pytest.mark.parametrize("key,ctr,expect", [
    ((0x0, 0x0), (0x0, 0x0), (0x6B200159, 0x99BA4EFE)),
    ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (0x1CB996FC, 0xBB002BE7)),
    ((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3), (0xC4923A9C, 0x483DF7A0)),
])
def test_threefry2x32_known_answers(key, ctr, expect):
    # Random123 kat_vectors (threefry2x32, 20 rounds); the same three cases as jax/tests/random_test.py
    x0, x1 = tf.threefry2x32(tf.ThreefryKey(*key), np.array([ctr[0]], dtype=np.uint32), np.array([ctr[1]], dtype=np.uint32))
    assert (int(x0[0]), int(x1[0])) == expect


def test_key_construction_and_split():
    k = tf.PRNGKey(42)
    assert list(map(int, k.k)) == [0, 42]
    assert list(map(int, tf.PRNGKey((7 << 32) + 5).k)) == [7, 5]
    a, b = tf.split(k)
    # key i of a split is the block function at counter (0, i)
    for i, child in enumerate((a, b)):
        x0, x1 = tf.threefry2x32(k, np.array([0], dtype=np.uint32), np.array([i], dtype=np.uint32))
        assert (int(child.k[0]), int(child.k[1])) == (int(x0[0]), int(x1[0]))
    many = tf.split(k, 5)
    assert len({tuple(map(int, c.k)) for c in many}) == 5
    assert tuple(map(int, many[1].k)) == tuple(map(int, b.k))  # prefix-stable
    k1, k2 = tf.get_keys(0)
    assert tuple(map(int, k1.k)) != tuple(map(int, k2.k))


def test_bits_uniform_normal_structure():
    k = tf.PRNGKey(3)
    b64 = tf.random_bits(k, 64, (4, 3))
    assert b64.dtype == np.uint64 and b64.shape == (4, 3)
    # row-major counters: the (i, j) element only depends on its linear index
    flat = tf.random_bits(k, 64, (12,))
    np.testing.assert_array_equal(b64.reshape(-1), flat)
    b32 = tf.random_bits(k, 32, (12,))
    np.testing.assert_array_equal(b32, (flat >> np.uint64(32)).astype(np.uint32) ^ (flat & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    u = tf.uniform(k, (20000,), np.float64)
    assert u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01
    z = tf.normal(k, (200000,))
    assert z.dtype == np.float64 and np.all(np.isfinite(z))
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs(np.mean(z ** 4) - 3.0) < 0.1
    z32 = tf.normal(k, (1000,), np.float32)
    assert z32.dtype == np.float32 and np.all(np.isfinite(z32))
    np.testing.assert_array_equal(tf.normal(k, (7, 5)), tf.normal(k, (7, 5)))  # counter-based: a pure function of the key


def test_predict_normals_layout():
    k = tf.get_keys(1)[1]
    e = tf.predict_normals(k, 4, 2, 6)
    assert e.shape == (4, 2, 6)
    ks = tf.split(k, 4)
    np.testing.assert_array_equal(e[2], tf.normal(ks[2], (2, 6)))


def test_models_accept_threefry_keys():
    from gpax_amd import ExactGP, _lib
    from tests.oracle_engine import OracleEngine
    _lib.set_engine(OracleEngine())
    try:
        rng = np.random.default_rng(0)
        X = np.linspace(0, 3, 15)
        y = np.sin(X) + 0.05 * rng.standard_normal(15)
        k1, k2 = tf.get_keys(0)
        m = ExactGP(1, "RBF")
        m.fit(k1, X, y, num_warmup=10, num_samples=10, progress_bar=False, print_summary=False)
        Xn = np.linspace(0, 3, 7)
        ym, ys = m.predict(k2, Xn, n=3)
        ym2, ys2 = m.predict(k2, Xn, n=3)
        np.testing.assert_array_equal(ys, ys2)
        # the draws are mean_s + L_s eps_s with eps_s = normal(split(k2, S)[s], (n, M))
        s = m.get_samples()
        eps = tf.predict_normals(k2, 10, 3, 7)
        p = {k: v[4] for k, v in s.items()}
        mean, cov = m.get_mvn_posterior(Xn, p)
        np.testing.assert_allclose(ys[4], mean[None, :] + eps[4] @ np.linalg.cholesky(cov).T, rtol=1e-7, atol=1e-9)
        # predict_in_batches: the same key for every slice
        yb, ysb = m.predict_in_batches(k2, Xn, batch_size=4, n=3)
        e0 = tf.predict_normals(k2, 10, 3, 4)
        mean0, cov0 = m.get_mvn_posterior(Xn[:4], p)
        np.testing.assert_allclose(ysb[4][:, :4], mean0[None, :] + e0[4] @ np.linalg.cholesky(cov0).T, rtol=1e-7, atol=1e-9)
    finally:
        _lib.set_engine(None)


def _ulps32(a, b):
    a, b = np.float32(a), np.float32(b)
    return abs(int(a.view(np.int32)) - int(b.view(np.int32)))


def test_partitionable_layout_matches_the_published_jax_values():
    """JAX >= 0.5 'Pseudorandom numbers' tutorial (jax_threefry_partitionable=True): key = random.key(42);
    random.normal(key) -> -0.028304616; then `for i in range(3): new_key, subkey = random.split(key); ...
    val = random.normal(subkey); key = new_key` prints 0.6057640314102173, -0.21089035272598267, -0.3948981463909149."""
    assert tf.set_partitionable(True) is True  # the default
    key = tf.PRNGKey(42)
    assert _ulps32(tf.normal(key, (), np.float32), -0.028304616) <= 4
    for expect in (0.6057640314102173, -0.21089035272598267, -0.3948981463909149):
        key, sub = tf.split(key)
        assert _ulps32(tf.normal(sub, (), np.float32), expect) <= 4


def test_legacy_layout_matches_the_published_jax_values():
    """JAX < 0.5 tutorial / README (jax_threefry_partitionable=False): split(PRNGKey(42)) ->
    [2465931498 3679230171], [255383827 267815257]; normal(PRNGKey(42)) -> -0.18471177, normal(subkey) -> 1.3694694;
    split(PRNGKey(0)) -> [4146024105 967050713], [2718843009 1272950319]; normal(PRNGKey(0)) -> -0.20584226."""
    old = tf.set_partitionable(False)
    try:
        k = tf.PRNGKey(42)
        new, sub = tf.split(k)
        assert tuple(map(int, new.k)) == (2465931498, 3679230171)
        assert tuple(map(int, sub.k)) == (255383827, 267815257)
        assert _ulps32(tf.normal(k, (), np.float32), -0.18471177) <= 4
        assert _ulps32(tf.normal(sub, (), np.float32), 1.3694694) <= 4
        k0 = tf.PRNGKey(0)
        a, b = tf.split(k0)
        assert tuple(map(int, a.k)) == (4146024105, 967050713)
        assert tuple(map(int, b.k)) == (2718843009, 1272950319)
        assert _ulps32(tf.normal(k0, (), np.float32), -0.20584226) <= 4
        # 64-bit draws consume two consecutive words per element; shapes only reshape the stream
        z = tf.normal(k0, (3, 2))
        assert z.shape == (3, 2) and np.all(np.isfinite(z))
        np.testing.assert_array_equal(z.reshape(-1), tf.normal(k0, (6,)))
    finally:
        tf.set_partitionable(old)
