"""NUTS through the model API with the transition loop in Python (GPX_NATIVE_NUTS=0) and in the library (default):
ExactGP(1, 'RBF').fit, 200 + 200, at the sizes the reference notebooks use and at C1's N = 512."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench_inputs
from gpax_amd import ExactGP
from gpax_amd.utils import get_keys

out = {}
for N in (25, 100, 128, 512):
    X, y, _, _ = bench_inputs.synthetic_problem(N, 1, 4, seed=1)
    rec = {}
    for native in ("0", "1", "0", "1"):
        os.environ["GPX_NATIVE_NUTS"] = native
        m = ExactGP(1, "RBF")
        t0 = time.perf_counter()
        m.fit(get_keys()[0], X, y, num_warmup=200, num_samples=200, progress_bar=False, print_summary=False)
        dt = time.perf_counter() - t0
        nl = int(np.sum(m.mcmc.get_extra_fields()[0]["n_leapfrog"]))
        rec.setdefault("native" if native == "1" else "python", []).append(dt)
        rec["leapfrogs_sampling"] = nl
    out[N] = {k: (min(v) if isinstance(v, list) else v) for k, v in rec.items()}
    print(N, out[N], flush=True)
print(json.dumps(out))
