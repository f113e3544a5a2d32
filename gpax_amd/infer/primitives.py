"""
`sample` / `plate` / `deterministic` — just enough of NumPyro's primitive vocabulary for the prior callables the
reference's models accept (gpax/models/gp.py:96-135: `kernel_prior`, `mean_fn_prior`, `noise_prior` are functions
that call `numpyro.sample` and return the sampled values, e.g. gpax/tests/test_gp.py:30-40).

NumPyro is not a dependency here and there is no tracing compiler: a prior callable is run ONCE, under `trace_sites`,
to read off its information content — which sites exist, their shapes (plates) and their distributions.  The log
joint and its gradient are then explicit (models/gp.py:_log_joint).  That only works when the callable returns the
sampled values themselves, under the site's own name, as the reference's callables do; a callable that post-processes
its draws (returns `2 * length`) has no MI355X path and is rejected with an explicit error.

    from gpax_amd import dist, sample, plate

    def gp_kernel_custom_prior():
        length = sample("k_length", dist.Uniform(0, 1))
        scale = sample("k_scale", dist.LogNormal(0, 1))
        return {"k_length": length, "k_scale": scale}
"""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import numpy as np

from . import dist as _dist

__all__ = ["sample", "plate", "deterministic", "trace_sites", "SiteValue", "seed"]

_STACK: list = []


class SiteValue(np.ndarray):
    """Placeholder returned by `sample` while a prior callable is being traced: the site's median, tagged with the
    site name.  Any arithmetic on it produces an untagged array, which is how post-processing is detected."""

    def __new__(cls, value, site: str):
        obj = np.asarray(value, dtype=np.float64).view(cls)
        obj._site = site
        return obj

    def __array_finalize__(self, obj):
        self._site = None  # views / ufunc results are not the pristine draw


class _Tracer:
    def __init__(self, rng=None):
        self.sites: List[Tuple[str, Tuple[int, ...], _dist.Distribution]] = []
        self.deterministic: Dict[str, float] = {}
        self.plates: List[Tuple[int, int]] = []  # (size, dim) of the active plates, dim < 0 as NumPyro counts them
        self.rng = rng  # seed(): sites return draws of their distribution instead of its median


class seed:
    """numpyro.handlers.seed(rng_seed=...) for prior programs run by hand: inside the block `sample` DRAWS from the
    site's distribution (a NumPy generator seeded with rng_seed) — what the reference's tests do with
    `m._sample_kernel_params()` / `m._sample_noise()` (gpax/tests/test_gp.py:79-127).  `.sites` lists what was registered."""

    def __init__(self, rng_seed=0):
        self._tr = _Tracer(np.random.default_rng(rng_seed))

    def __enter__(self):
        _STACK.append(self._tr)
        return self

    def __exit__(self, *exc):
        _STACK.pop()
        return False

    @property
    def sites(self):
        return list(self._tr.sites)


def _plate_shape(plates) -> Tuple[int, ...]:
    """Batch shape of a site under the active plates, by NumPyro's rule: a plate occupies the batch dimension `dim`
    it names (negative, from the right) or else the first free one counting -1, -2, ... in the order the plates
    were entered — so of two nested plates without `dim` the OUTER one is the last axis."""
    taken = {}
    for size, dim in plates:
        taken[dim] = size
    if not taken:
        return ()
    nd = -min(taken)
    return tuple(taken.get(-nd + i, 1) for i in range(nd))


def sample(name: str, fn, obs=None, **kwargs):
    """numpyro.sample(name, dist): registers the site with the active trace and returns a tagged placeholder."""
    if not _STACK:
        raise RuntimeError("gpax_amd.sample() is only meaningful inside a prior callable passed to a model "
                           "(kernel_prior / mean_fn_prior / noise_prior); models trace it once to learn the sites")
    if not isinstance(fn, _dist.Distribution):
        raise NotImplementedError(f"site {name!r}: expected a gpax_amd.dist distribution, got {type(fn).__name__}")
    if obs is not None:
        raise NotImplementedError("observed sites inside prior callables have no MI355X path")
    tr = _STACK[-1]
    if any(s[0] == name for s in tr.sites):
        raise ValueError(f"site {name!r} sampled twice")
    shape = _plate_shape(tr.plates)
    tr.sites.append((name, shape, fn))
    if tr.rng is not None:  # under seed(): a draw
        return SiteValue(np.asarray(fn.sample(tr.rng, shape), dtype=np.float64), name)
    med = np.full(shape, float(fn.median())) if shape else float(fn.median())
    return SiteValue(med, name)


def deterministic(name: str, value):
    """numpyro.deterministic(name, value): a fixed value that shows up among the samples."""
    if _STACK:
        _STACK[-1].deterministic[name] = value
    return value


class plate:
    """numpyro.plate(name, size, dim=None): sites sampled inside get a batch dimension of `size` (gp.py:238-239
    'ard'; nested with explicit dims in vgp.py:104-105: dim=-2 tasks, dim=-1 lengthscales)."""

    def __init__(self, name: str, size: int, subsample_size=None, dim=None):
        if subsample_size is not None:
            raise NotImplementedError("plate(subsample_size=...) has no MI355X path")
        if dim is not None and int(dim) >= 0:
            raise ValueError("plate dim must be negative (counted from the right), as in NumPyro")
        self.name, self.size, self.dim = name, int(size), (None if dim is None else int(dim))

    def __enter__(self):
        if _STACK:
            active = _STACK[-1].plates
            used = {d for _, d in active}
            dim = self.dim
            if dim is None:
                dim = -1
                while dim in used:
                    dim -= 1
            elif dim in used:
                raise ValueError(f"plate {self.name!r}: batch dimension {dim} is already taken by an enclosing plate")
            active.append((self.size, dim))
        return self

    def __exit__(self, *exc):
        if _STACK:
            _STACK[-1].plates.pop()
        return False


def trace_sites(fn: Callable, what: str):
    """Run the prior callable once and return (sites, returned) where sites = [(name, shape, distribution)] and
    `returned` is what the callable returned (a dict key -> SiteValue, or a single SiteValue for noise_prior).
    Also accepts the declarative form: a dict name -> distribution (or a callable returning one)."""
    if isinstance(fn, dict):
        spec = fn
    else:
        tr = _Tracer()
        _STACK.append(tr)
        try:
            spec = fn()
        finally:
            _STACK.pop()
        if not (isinstance(spec, dict) and spec and all(isinstance(v, _dist.Distribution) for v in spec.values())):
            return _check_returned(tr, spec, what)
    if not all(isinstance(v, _dist.Distribution) for v in spec.values()):
        raise NotImplementedError(f"{what}: a dict must map site names to gpax_amd.dist distributions")
    sites = [(k, (), v) for k, v in spec.items()]
    return sites, {k: SiteValue(float(v.median()), k) for k, v in spec.items()}, {}


def _check_returned(tr: _Tracer, ret, what: str):
    if not tr.sites:
        raise NotImplementedError(f"{what} registered no sites: use gpax_amd.sample(name, dist) inside it "
                                  "(numpyro.sample has no MI355X path) or return a dict name -> distribution")
    items = ret.items() if isinstance(ret, dict) else [(None, ret)]
    names = {s[0] for s in tr.sites}
    for key, val in items:
        if isinstance(val, SiteValue) and val._site is not None:
            if key is not None and val._site != key:
                raise NotImplementedError(f"{what}: returns site {val._site!r} under the key {key!r}; return each "
                                          "sampled value under its own site name")
            continue
        if key is not None and key in tr.deterministic:
            continue
        if val is None:  # the reference's default returns period=None for non-periodic kernels (gp.py:245)
            continue
        raise NotImplementedError(f"{what}: the value returned for {key!r} is not the sampled site itself "
                                  "(post-processed draws cannot be differentiated without a tracing compiler); "
                                  f"sites seen: {sorted(names)}")
    return tr.sites, ret, tr.deterministic
