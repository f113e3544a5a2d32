"""
Sample-sharded predictive sweep across processes, on top of ANY communicator.

The only axis of the exact-GP path that shards is the vmap over posterior samples in
ExactGP.predict (gpax/models/gp.py:393-395): the S per-theta pipelines are independent given
(X_train, y_train, X_new), and `y_means.mean(0)` (gp.py:399) is the only cross-sample reduction.

The product's own multi-process path is `_lib.Rank` (include/gpx.h gpx_rank_*: RCCL broadcast / gather inside the
library, rendezvous in gpax_amd/launch.py) — `ExactGP.predict_distributed()` uses it by default and nothing here is
involved.  This module is the host-level variant for callers that already live inside another runtime's process
group (mpi4py, torch.distributed, ...): `predict_sharded` only needs an object with

    comm.rank, comm.world
    comm.bcast(array_or_None) -> array          (shape / dtype known on rank 0 only)
    comm.gather_rows(local, counts) -> array on rank 0, None elsewhere

and imports no such runtime itself (tools/torch_comm.py holds a torch.distributed implementation; the CPU tests drive
it over gloo with world sizes 2 and 3).

    rank 0 broadcasts (X, y_res, X_new, theta table, eps)      [KBs .. a few MB]
    rank r sweeps its contiguous block of samples on its own GPU (gpx_predict_sweep)
    results are gathered on rank 0                              [S*(n+1)*M doubles]
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np


def shard_range(S: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of rank `rank` out of S samples, sizes differing by <= 1
    (the rule of gpx_shard_range)."""
    base, rem = divmod(S, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_ranges_weighted(S: int, weights) -> list:
    """All blocks [(start, stop), ...] over S samples with sizes in proportion to `weights` (largest-remainder rounding,
    ties to the lower rank) — the rule of gpx_shard_ranges_weighted, which sizes the ranks' blocks of the library's own
    sweep by measured GPU speed (gpx_rank_calibrate).  Weights that are not all positive and finite: equal blocks."""
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    world = int(w.size)
    if not (np.all(w > 0.0) and np.all(w < 1e300)):
        return [shard_range(S, r, world) for r in range(world)]
    tot = 0.0
    for v in w:  # (summed in rank order, as the library does)
        tot += float(v)
    share = [S * (float(v) / tot) for v in w]
    cnt = [int(x) for x in share]
    frac = [x - c for x, c in zip(share, cnt)]
    for _ in range(S - sum(cnt)):
        best = 0
        for r in range(1, world):
            if frac[r] > frac[best]:
                best = r
        cnt[best] += 1
        frac[best] = -1.0
    out, at = [], 0
    for c in cnt:
        out.append((at, at + c))
        at += c
    return out


def predict_sharded(engine, kind: int, X, yres, Xnew, samples: Optional[Dict[str, np.ndarray]], eps,
                    noiseless: bool, jitter: float, comm, weights=None):
    """The S-sample predictive sweep, sharded over comm.world ranks (`engine`: one Engine or a list of
    contexts on this rank's GPU, see _lib.get_sweep_engines).  Inputs need only be valid on rank 0.  weights (rank 0;
    one positive number per rank, e.g. measured GPU speeds): block sizes in proportion, else equal blocks.  Returns (means (S, M), y_sampled (S, n, M), infos (S,)) on rank 0, None elsewhere."""
    X = comm.bcast(X)
    yres = comm.bcast(yres)
    Xnew = comm.bcast(Xnew)
    ells = comm.bcast(None if samples is None else np.asarray(samples["k_length"], dtype=np.float64))
    scales = comm.bcast(None if samples is None else np.asarray(samples["k_scale"], dtype=np.float64))
    noises = comm.bcast(None if samples is None else np.asarray(samples["noise"], dtype=np.float64))
    eps = comm.bcast(eps)
    S = ells.shape[0]
    w = comm.bcast(np.ones(comm.world) if weights is None else np.asarray(weights, dtype=np.float64).reshape(-1))
    blocks = shard_ranges_weighted(S, w) if w.size == comm.world else [shard_range(S, r, comm.world) for r in range(comm.world)]
    lo, hi = blocks[comm.rank]
    counts = [b - a for a, b in blocks]
    M, n = Xnew.shape[0], eps.shape[1]
    if hi > lo:
        from ._lib import concurrent_sweep

        engines = engine if isinstance(engine, (list, tuple)) else [engine]
        yr = yres if yres.ndim == 1 else yres[lo:hi]
        means, draws, infos = concurrent_sweep(list(engines), X, kind, ells[lo:hi].reshape(hi - lo, -1), scales[lo:hi],
                                               noises[lo:hi], yr, Xnew, noiseless, jitter, eps[lo:hi])
    else:
        means, draws, infos = np.empty((0, M)), np.empty((0, n, M)), np.empty((0,), dtype=np.int32)
    means = comm.gather_rows(means, counts)
    draws = comm.gather_rows(draws, counts)
    infos = comm.gather_rows(np.asarray(infos, dtype=np.int32), counts)
    if comm.rank != 0:
        return None
    return means, draws, infos
