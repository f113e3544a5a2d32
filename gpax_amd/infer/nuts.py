"""
Host-side No-U-Turn sampler (multinomial NUTS with dual-averaging step size and windowed
diagonal mass-matrix adaptation) driving a device potential.

Replaces numpyro.infer.{NUTS, MCMC} as used by ExactGP.fit (gpax/models/gp.py:207-218):
NUTS defaults target_accept_prob=0.8, max_tree_depth=10, diagonal mass matrix, Stan-style
warm-up windows.  Every leapfrog calls `potential_and_grad(u)`, which for the exact GP is one
Gram + Cholesky + K^-1 + gradient-contraction pass on the GPU (gpx_factor + gpx_lml_grad).
JAX's threefry streams cannot be reproduced, so chains are not bit-comparable with NumPyro's;
the sampler is validated on distributional properties (tests/test_samplers.py) and its deterministic building
blocks on known answers (tests/test_sampler_known_answers.py).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Tuple

import numpy as np

MAX_DELTA_ENERGY = 1000.0


class _DualAveraging:
    def __init__(self, eps0: float, target: float = 0.8, t0: float = 10.0, kappa: float = 0.75, gamma: float = 0.05):
        self.mu = math.log(10.0 * eps0)
        self.target, self.t0, self.kappa, self.gamma = target, t0, kappa, gamma
        self.t = 0
        self.h_bar = 0.0
        self.log_eps = math.log(eps0)
        self.log_eps_bar = 0.0

    def update(self, accept_prob: float):
        self.t += 1
        w = 1.0 / (self.t + self.t0)
        self.h_bar = (1 - w) * self.h_bar + w * (self.target - accept_prob)
        self.log_eps = self.mu - math.sqrt(self.t) / self.gamma * self.h_bar
        eta = self.t ** (-self.kappa)
        self.log_eps_bar = eta * self.log_eps + (1 - eta) * self.log_eps_bar

    def restart(self, eps: float):
        self.__init__(eps, self.target, self.t0, self.kappa, self.gamma)


class _Welford:
    def __init__(self, dim):
        self.n = 0
        self.mean = np.zeros(dim)
        self.m2 = np.zeros(dim)

    def update(self, x):
        self.n += 1
        d = x - self.mean
        self.mean += d / self.n
        self.m2 += d * (x - self.mean)

    def variance(self):
        var = self.m2 / max(self.n - 1, 1)
        n = self.n
        return (n / (n + 5.0)) * var + 1e-3 * (5.0 / (n + 5.0))  # Stan / NumPyro regularisation


def adaptation_schedule(num_warmup: int):
    """Stan-style windows (NumPyro build_adaptation_schedule): list of (start, end) inclusive."""
    if num_warmup < 20:
        return [(0, num_warmup - 1)]
    init_buf, term_buf, base = 75, 50, 25
    if init_buf + base + term_buf > num_warmup:
        init_buf = int(0.15 * num_warmup)
        term_buf = int(0.1 * num_warmup)
        base = num_warmup - init_buf - term_buf
    sched = [(0, init_buf - 1)]
    end_slow = num_warmup - term_buf
    start, size = init_buf, base
    while start < end_slow:
        end = start + size
        if end + 2 * size > end_slow:  # absorb the remainder into this window
            end = end_slow
        sched.append((start, end - 1))
        start, size = end, 2 * size
    sched.append((end_slow, num_warmup - 1))
    return sched


def _leapfrog(pe_fn, u, p, g, eps, inv_mass):
    p = p - 0.5 * eps * g
    u = u + eps * inv_mass * p
    U, g = pe_fn(u)
    p = p - 0.5 * eps * g
    return u, p, U, g


def _energy(U, p, inv_mass):
    if not np.isfinite(U):
        return np.inf
    return U + 0.5 * float(p @ (inv_mass * p))


def _uturn(rho, p_left, p_right, inv_mass):
    """Generalised termination criterion (Betancourt 2017, A.4.2) in the form NumPyro's `_is_turning` uses: the summed
    momentum of the trajectory with its two end points weighted one half, against the velocities at both ends."""
    rho = rho - 0.5 * (p_left + p_right)
    return (rho @ (inv_mass * p_left) <= 0) or (rho @ (inv_mass * p_right) <= 0)


def _build_tree(pe_fn, u, p, g, direction, depth, eps, inv_mass, H0, rng):
    """Recursive doubling.  Returns a dict describing the subtree."""
    if depth == 0:
        u1, p1, U1, g1 = _leapfrog(pe_fn, u, p, g, direction * eps, inv_mass)
        H1 = _energy(U1, p1, inv_mass)
        dH = H1 - H0
        if np.isnan(dH):
            dH = np.inf
        diverging = dH > MAX_DELTA_ENERGY
        return dict(ul=u1, pl=p1, gl=g1, ur=u1, pr=p1, gr=g1, prop=(u1, U1, g1), logw=-dH, rho=p1.copy(),
                    turning=False, diverging=diverging, sum_accept=min(1.0, math.exp(min(0.0, -dH))), n=1)
    a = _build_tree(pe_fn, u, p, g, direction, depth - 1, eps, inv_mass, H0, rng)
    if a["turning"] or a["diverging"]:
        return a
    if direction == 1:
        b = _build_tree(pe_fn, a["ur"], a["pr"], a["gr"], direction, depth - 1, eps, inv_mass, H0, rng)
    else:
        b = _build_tree(pe_fn, a["ul"], a["pl"], a["gl"], direction, depth - 1, eps, inv_mass, H0, rng)
    logw = np.logaddexp(a["logw"], b["logw"])
    prop = a["prop"]
    if not (b["turning"] or b["diverging"]):
        if math.log(rng.uniform()) < b["logw"] - logw:  # multinomial within the new subtree pair
            prop = b["prop"]
    rho = a["rho"] + b["rho"]
    if direction == 1:
        ul, pl, gl, ur, pr, gr = a["ul"], a["pl"], a["gl"], b["ur"], b["pr"], b["gr"]
    else:
        ul, pl, gl, ur, pr, gr = b["ul"], b["pl"], b["gl"], a["ur"], a["pr"], a["gr"]
    turning = b["turning"] or _uturn(rho, pl, pr, inv_mass)
    return dict(ul=ul, pl=pl, gl=gl, ur=ur, pr=pr, gr=gr, prop=prop, logw=logw, rho=rho, turning=turning,
                diverging=b["diverging"], sum_accept=a["sum_accept"] + b["sum_accept"], n=a["n"] + b["n"])


def nuts_transition(pe_fn, u, U, g, eps, inv_mass, rng, max_tree_depth=10):
    """One NUTS transition.  Returns (u, U, g, mean_accept_prob, n_leapfrog, diverging)."""
    dim = u.shape[0]
    p0 = rng.standard_normal(dim) / np.sqrt(inv_mass)
    H0 = _energy(U, p0, inv_mass)
    ul = ur = u
    pl = pr = p0
    gl = gr = g
    prop = (u, U, g)
    logw = 0.0
    rho = p0.copy()
    sum_accept, n_leap = 0.0, 0
    diverging = False
    for depth in range(max_tree_depth):
        direction = 1 if rng.uniform() < 0.5 else -1
        if direction == 1:
            t = _build_tree(pe_fn, ur, pr, gr, 1, depth, eps, inv_mass, H0, rng)
            ur, pr, gr = t["ur"], t["pr"], t["gr"]
        else:
            t = _build_tree(pe_fn, ul, pl, gl, -1, depth, eps, inv_mass, H0, rng)
            ul, pl, gl = t["ul"], t["pl"], t["gl"]
        sum_accept += t["sum_accept"]
        n_leap += t["n"]
        if t["diverging"]:
            diverging = True
            break
        if t["turning"]:
            break
        if math.log(rng.uniform()) < t["logw"] - logw:  # biased progressive sampling
            prop = t["prop"]
        logw = np.logaddexp(logw, t["logw"])
        rho = rho + t["rho"]
        if _uturn(rho, pl, pr, inv_mass):
            break
    return prop[0], prop[1], prop[2], sum_accept / max(n_leap, 1), n_leap, diverging


def find_reasonable_step_size(pe_fn, u, U, g, inv_mass, rng, eps=1.0):
    """Heuristic of Hoffman & Gelman (Alg. 4): double/halve until accept prob crosses 0.8."""
    p = rng.standard_normal(u.shape[0]) / np.sqrt(inv_mass)
    H0 = _energy(U, p, inv_mass)
    _, p1, U1, _ = _leapfrog(pe_fn, u, p, g, eps, inv_mass)
    dH = H0 - _energy(U1, p1, inv_mass)
    direction = 1 if (np.isfinite(dH) and dH > math.log(0.8)) else -1
    for _ in range(50):
        eps = eps * (2.0 ** direction)
        _, p1, U1, _ = _leapfrog(pe_fn, u, p, g, eps, inv_mass)
        dH = H0 - _energy(U1, p1, inv_mass)
        ok = np.isfinite(dH) and dH > math.log(0.8)
        if (direction == 1 and not ok) or (direction == -1 and ok):
            break
    return eps


def run_nuts(potential_and_grad: Callable[[np.ndarray], Tuple[float, np.ndarray]], u0: np.ndarray,
             num_warmup: int, num_samples: int, rng: np.random.Generator, target_accept: float = 0.8,
             max_tree_depth: int = 10, progress: Callable[[int, int, dict], None] = None,
             transition: Callable = None) -> Dict[str, np.ndarray]:
    """Warm-up + sampling for one chain.  Returns unconstrained draws and diagnostics.
    transition(u, U, g, eps, inv_mass, rng, max_tree_depth): a drop-in for nuts_transition over the same potential — the
    library's own loop (gpx_nuts_transition, csrc/nuts.hip), which consumes `rng` exactly as nuts_transition does."""
    step = nuts_transition if transition is None else (lambda pe, *a: transition(*a))
    u = np.array(u0, dtype=np.float64)
    dim = u.shape[0]
    U, g = potential_and_grad(u)
    if not np.isfinite(U):
        raise FloatingPointError("NUTS: non-finite potential at the initial point")
    inv_mass = np.ones(dim)
    eps = find_reasonable_step_size(potential_and_grad, u, U, g, inv_mass, rng, 1.0) if num_warmup > 0 else 1.0
    da = _DualAveraging(eps, target_accept)
    sched = adaptation_schedule(num_warmup) if num_warmup > 0 else []
    window = 0
    wf = _Welford(dim)
    draws = np.empty((num_samples, dim))
    stats = dict(accept=np.empty(num_samples), n_leapfrog=np.empty(num_samples, dtype=int),
                 diverging=np.zeros(num_samples, dtype=bool), potential=np.empty(num_samples))
    total = num_warmup + num_samples
    for it in range(total):
        warm = it < num_warmup
        u, U, g, acc, nl, div = step(potential_and_grad, u, U, g, eps, inv_mass, rng, max_tree_depth)
        if warm:
            da.update(acc)
            eps = math.exp(da.log_eps)
            in_slow = 0 < window < len(sched) - 1
            if in_slow:
                wf.update(u)
            if it == sched[window][1]:
                if in_slow and wf.n > 1:
                    inv_mass = wf.variance()
                    wf = _Welford(dim)
                    eps = find_reasonable_step_size(potential_and_grad, u, U, g, inv_mass, rng, eps)  # from the current step
                    da.restart(eps)
                window += 1
            if it == num_warmup - 1:
                eps = math.exp(da.log_eps_bar)
        else:
            k = it - num_warmup
            draws[k] = u
            stats["accept"][k] = acc
            stats["n_leapfrog"][k] = nl
            stats["diverging"][k] = div
            stats["potential"][k] = U
        if progress is not None:
            progress(it, total, dict(step_size=eps, n_leapfrog=nl, accept=acc, warmup=warm))
    stats["step_size"] = eps
    stats["inv_mass"] = inv_mass
    return dict(draws=draws, **stats)
