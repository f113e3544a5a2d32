"""
JAX-compatible counter-based PRNG on the host: threefry2x32 keys, `split` and `normal`
as `jax.random` defines them, so that the standard normals behind `ExactGP.predict` can be the ones a JAX run of
the reference would draw from the same key (gpax/utils/utils.py:24-30 get_keys; gpax/models/gp.py:292,393:
`keys = jax.random.split(rng_key, S)`, per sample `MultivariateNormal(mean, K).sample(key_s, (n,))`
= `mean + L @ jax.random.normal(key_s, (n, M))` in NumPyro).

What is pinned (tests/test_threefry.py).  The threefry2x32 block function: the Random123 / JAX known-answer
vectors.  The layers above it — key construction, `split`, `random_bits`, `uniform`, `normal` — restate
jax/_src/prng.py and jax/_src/random.py [knowledge] in BOTH counter layouts JAX has shipped, and are pinned to the
values JAX's own documentation publishes for them:
  * `jax_threefry_partitionable=True` (default since JAX 0.5.0; the reference pins jax >= 0.6.2 — the default
    here): "Pseudorandom numbers" tutorial, JAX >= 0.5: normal(key(42)) = -0.028304616 and the three draws
    0.6057640314102173, -0.21089035272598267, -0.3948981463909149 of its split loop;
  * the legacy layout (`set_partitionable(False)`): the same tutorial before 0.5 and the README / quickstart:
    split(PRNGKey(42)) = [2465931498 3679230171], [255383827 267815257], normal = -0.18471177 / 1.3694694;
    split(PRNGKey(0)) = [4146024105 967050713], [2718843009 1272950319], normal(PRNGKey(0)) = -0.20584226.
The uint32 words are reproduced exactly; `normal` ends in erf_inv, where XLA's polynomial and
scipy.special.erfinv agree to a few ulp of float32, not bit for bit (the tests allow 4 ulp).
"""
from __future__ import annotations

import numpy as np
from scipy import special

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
_U32 = np.uint32
_PARTITIONABLE = True


def set_partitionable(flag: bool) -> bool:
    """jax.config.update('jax_threefry_partitionable', flag): choose the counter layout of split / random_bits.
    Returns the previous setting."""
    global _PARTITIONABLE
    old, _PARTITIONABLE = _PARTITIONABLE, bool(flag)
    return old


class ThreefryKey:
    """A raw threefry key: two uint32 words (jax.random.PRNGKey(seed) = [seed >> 32, seed & 0xffffffff])."""

    __slots__ = ("k",)

    def __init__(self, k1, k2):
        self.k = np.array([k1, k2], dtype=np.uint32)

    def __repr__(self):
        return f"ThreefryKey([{int(self.k[0])}, {int(self.k[1])}])"

    def __iter__(self):
        return iter(self.k)


def PRNGKey(seed: int) -> ThreefryKey:
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return ThreefryKey(seed >> 32, seed & 0xFFFFFFFF)


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(key, x0, x1):
    """The 20-round Threefry-2x32 block function on arrays of counters (x0, x1); returns two uint32 arrays."""
    k1, k2 = (np.uint32(v) for v in (key.k if isinstance(key, ThreefryKey) else key))
    with np.errstate(over="ignore"):
        ks = (k1, k2, k1 ^ k2 ^ _U32(0x1BD11BDA))
        x0 = np.asarray(x0, dtype=np.uint32) + ks[0]
        x1 = np.asarray(x1, dtype=np.uint32) + ks[1]
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(i + 1) % 3]
            x1 = x1 + ks[(i + 2) % 3] + _U32(i + 1)
    return x0, x1


def _iota_2x32(shape):
    """High and low 32-bit words of the row-major linear index of every element of `shape`."""
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64).reshape(shape)
    return (idx >> np.uint64(32)).astype(np.uint32), (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def _legacy_bits32(key: ThreefryKey, n: int) -> np.ndarray:
    """threefry_2x32(key, iota(n)) of the non-partitionable layout: the counter vector (padded to even length) is cut
    in two halves that feed the two input words; the output words are concatenated."""
    cnt = np.arange(n, dtype=np.uint32)
    if n % 2:
        cnt = np.concatenate([cnt, np.zeros(1, dtype=np.uint32)])
    h = cnt.size // 2
    a, b = threefry2x32(key, cnt[:h], cnt[h:])
    return np.concatenate([a, b])[:n]


def split(key: ThreefryKey, num: int = 2):
    """jax.random.split.  Partitionable layout: key i = threefry2x32(key, (0, i)).  Legacy layout: the 2 num words of
    threefry_2x32(key, iota(2 num)) reshaped to (num, 2)."""
    if not _PARTITIONABLE:
        w = _legacy_bits32(key, 2 * int(num)).reshape(int(num), 2)
        return [ThreefryKey(a, b) for a, b in w]
    hi, lo = _iota_2x32((int(num),))
    b1, b2 = threefry2x32(key, hi, lo)
    return [ThreefryKey(a, b) for a, b in zip(b1, b2)]


def random_bits(key: ThreefryKey, bit_width: int, shape):
    shape = tuple(shape)
    if not _PARTITIONABLE:
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if bit_width == 32:
            return _legacy_bits32(key, n).reshape(shape)
        if bit_width == 64:  # two 32-bit words per element, consecutive in the stream: (hi << 32) | lo
            w = _legacy_bits32(key, 2 * n).reshape(n, 2).astype(np.uint64)
            return ((w[:, 0] << np.uint64(32)) | w[:, 1]).reshape(shape)
        raise NotImplementedError("bit_width must be 32 or 64")
    hi, lo = _iota_2x32(shape)
    b1, b2 = threefry2x32(key, hi, lo)
    if bit_width == 64:
        return (b1.astype(np.uint64) << np.uint64(32)) | b2.astype(np.uint64)
    if bit_width == 32:
        return b1 ^ b2
    raise NotImplementedError("bit_width must be 32 or 64")


def uniform(key: ThreefryKey, shape, dtype=np.float64, minval=0.0, maxval=1.0):
    """jax.random.uniform: mantissa bits of a float in [1, 2) minus 1, scaled to [minval, maxval)."""
    dtype = np.dtype(dtype)
    nbits, nmant, utype = (64, 52, np.uint64) if dtype == np.float64 else (32, 23, np.uint32)
    bits = random_bits(key, nbits, shape)
    one = np.array(1.0, dtype).view(utype)
    floats = ((bits >> utype(nbits - nmant)) | one).view(dtype) - dtype.type(1.0)
    minval, maxval = dtype.type(minval), dtype.type(maxval)
    return np.maximum(minval, floats * (maxval - minval) + minval)


def normal(key: ThreefryKey, shape, dtype=np.float64):
    """jax.random.normal: sqrt(2) * erf_inv(u), u uniform on (-1, 1)."""
    dtype = np.dtype(dtype)
    lo = np.nextafter(dtype.type(-1.0), dtype.type(0.0))
    u = uniform(key, shape, dtype, lo, dtype.type(1.0))
    return (dtype.type(np.sqrt(2.0)) * special.erfinv(u)).astype(dtype)


def get_keys(seed: int = 0):
    """gpax.utils.get_keys (utils.py:24-30): `split(PRNGKey(seed))` -> one key for inference, one for prediction."""
    k1, k2 = split(PRNGKey(seed))
    return k1, k2


def predict_normals(key: ThreefryKey, S: int, n: int, M: int, dtype=np.float64) -> np.ndarray:
    """The (S, n, M) standard normals of ExactGP.predict for this key (gp.py:393 split over the S samples,
    gp.py:292 MultivariateNormal.sample(key_s, (n,)) -> normal(key_s, (n, M)))."""
    return np.stack([normal(k, (n, M), dtype) for k in split(key, S)]) if S else np.empty((0, n, M), dtype)
