"""
Sample-sharded predictive sweep across the GPUs of one node.

The only axis of the exact-GP path that shards is the vmap over posterior samples in
ExactGP.predict (gpax/models/gp.py:393-395): the S per-theta pipelines are independent given
(X_train, y_train, X_new), and `y_means.mean(0)` (gp.py:399) is the only cross-sample reduction.
One process per GPU (torch.distributed.run), one libgpx context per process:

    rank 0 broadcasts (X, y_res, X_new, theta table, eps)      [KBs .. a few MB]
    rank r sweeps its contiguous block of samples on its own GPU (gpx_predict_sweep)
    results are gathered on rank 0                              [S*(n+1)*M doubles]

No collective sits inside the sweep.  torch.distributed is launcher / transport plumbing only
(backend "nccl" = RCCL over xGMI with GPU staging tensors; "gloo" on CPU in the tests); the
compute never touches torch.  Import torch BEFORE gpax_amd in such a process so that libgpx binds
to the HIP runtime torch already loaded (same SONAME, one runtime per process).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np


def shard_range(S: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of rank `rank` out of S samples, sizes differing by <= 1."""
    base, rem = divmod(S, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class Communicator:
    """Minimal array collectives on top of an initialised torch.distributed process group."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if device is None:
            if dist.get_backend() == "nccl":
                device = torch.device("cuda", torch.cuda.current_device())
            else:
                device = torch.device("cpu")
        self.device = device

    def bcast(self, arr: Optional[np.ndarray], src: int = 0) -> np.ndarray:
        """Broadcast an array whose shape / dtype are only known on `src`."""
        torch, dist = self.torch, self.dist
        meta = [None]
        if self.rank == src:
            arr = np.ascontiguousarray(arr)
            if not arr.flags.writeable:  # torch.from_numpy wants a writable buffer (e.g. broadcast views)
                arr = arr.copy()
            meta = [(tuple(arr.shape), str(arr.dtype))]
        dist.broadcast_object_list(meta, src=src)
        shape, dtype = meta[0]
        if self.rank == src:
            t = torch.from_numpy(arr).to(self.device)
        else:
            t = torch.empty(shape, dtype=getattr(torch, dtype), device=self.device)
        dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def gather_rows(self, local: np.ndarray, counts) -> Optional[np.ndarray]:
        """Concatenate per-rank blocks (leading axis, `counts[r]` rows from rank r) on rank 0."""
        torch, dist = self.torch, self.dist
        local = np.ascontiguousarray(local)
        maxc = max(max(counts), 1)
        pad = np.zeros((maxc,) + local.shape[1:], dtype=local.dtype)
        pad[: local.shape[0]] = local
        t = torch.from_numpy(pad).to(self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        if self.rank != 0:
            return None
        return np.concatenate([o.cpu().numpy()[: counts[r]] for r, o in enumerate(out)], axis=0)

    def barrier(self):
        self.dist.barrier()


def predict_sharded(engine, kind: int, X, yres, Xnew, samples: Optional[Dict[str, np.ndarray]], eps,
                    noiseless: bool, jitter: float, comm: Communicator):
    """The S-sample predictive sweep, sharded over comm.world ranks (`engine`: one Engine or a list of
    contexts on this rank's GPU, see _lib.get_sweep_engines).  Inputs need only be valid on rank 0.  Returns (means (S, M), y_sampled (S, n, M), infos (S,)) on rank 0, None elsewhere."""
    X = comm.bcast(X)
    yres = comm.bcast(yres)
    Xnew = comm.bcast(Xnew)
    ells = comm.bcast(None if samples is None else np.asarray(samples["k_length"], dtype=np.float64))
    scales = comm.bcast(None if samples is None else np.asarray(samples["k_scale"], dtype=np.float64))
    noises = comm.bcast(None if samples is None else np.asarray(samples["noise"], dtype=np.float64))
    eps = comm.bcast(eps)
    S = ells.shape[0]
    lo, hi = shard_range(S, comm.rank, comm.world)
    counts = [shard_range(S, r, comm.world)[1] - shard_range(S, r, comm.world)[0] for r in range(comm.world)]
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
