"""torch.distributed implementation of the communicator protocol `gpax_amd.parallel.predict_sharded` takes (rank, world,
bcast, gather_rows) — NOT part of the product package (gpax_amd imports no torch): for callers who already run inside a
torch process group, and for the CPU tests, which drive the N > 1 control flow over gloo (tests/test_parallel_gloo.py).
Backend "nccl" = RCCL over xGMI with GPU staging tensors; import torch BEFORE gpax_amd in such a process so that libgpx
binds to the HIP runtime torch already loaded."""
from typing import Optional

import numpy as np


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
