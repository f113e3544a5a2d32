"""Run under `python -m torch.distributed.run --nproc-per-node N`: ExactGP.predict_distributed over the nccl (RCCL)
backend equals the single-process predict.  (1 GPU box: N = 1; the same code path runs at N = 8.)"""
import os, sys
import torch  # first: libgpx must bind to the HIP runtime torch loads
import torch.distributed as dist
import numpy as np
sys.path.insert(0, os.getcwd())
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
from gpax_amd import ExactGP
from gpax_amd.utils import get_keys
from oracle import cpu_ref as ref
rank = dist.get_rank()
m = ExactGP(2, "Matern")
m._device = local
if rank == 0:
    X, y, Xn, p = ref.synthetic_problem(700, 2, 130, seed=0)
    th = ref.synthetic_theta_samples(37, 2, seed=1)
    m.X_train, m.y_train = m._set_data(X, y)
    res = m.predict_distributed(get_keys()[1], Xn, th, n=2)
    one = m.predict(get_keys()[1], Xn, th, n=2)
    err = max(np.abs(res[0] - one[0]).max(), np.abs(res[1] - one[1]).max())
    print(f"world={dist.get_world_size()} predict_distributed vs predict: max abs diff {err:.3e}", flush=True)
    assert err == 0.0
else:
    assert m.predict_distributed(None, None) is None
dist.barrier()
dist.destroy_process_group()
