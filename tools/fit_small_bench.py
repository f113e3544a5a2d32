"""The fused small-N fit step (csrc/fit_small.hip: Gram + factorisation + lml + gradient in ONE launch, N <= 128) against
the general launch sequence (GPX_FIT_SMALL=0): device time per launch (HIP events around the kernel), host time per
gpx_factor + gpx_lml_grad pair and per gpx_fit_batch call at B = 1 / 4 / 64.  One JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import bench_inputs  # noqa: E402
from gpax_amd import _lib  # noqa: E402

out = []
for N, d, kind in [(7, 1, 0), (25, 1, 0), (40, 2, 1), (64, 2, 0), (100, 2, 1), (127, 3, 0), (128, 2, 0)]:
    X, y, _, p = bench_inputs.synthetic_problem(N, d, 4, seed=N)
    rec = {"N": N, "d": d, "kernel": ["RBF", "Matern"][kind]}
    for mode in ("1", "0"):
        os.environ["GPX_FIT_SMALL"] = mode
        e = _lib.Engine(0)
        e.set_train(X)

        def pair():
            e.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
            e.lml_grad()

        for _ in range(50):
            pair()
        t0 = time.perf_counter()
        for _ in range(400):
            pair()
        host_pair = (time.perf_counter() - t0) / 400
        e.profile_enable(True)
        e.profile_reset()
        for _ in range(50):
            pair()
        dev = {}
        for name, cls in (("gemm_other", _lib.PROF_GEMM_OTHER), ("potf2", _lib.PROF_POTF2), ("gram", _lib.PROF_GRAM)):
            n, ms, _ = e.profile_read(cls)
            dev[name] = [n / 50, ms / 50 * 1e3]
        e.profile_enable(False)
        fb = {}
        for B in (1, 4, 64):
            ells = np.stack([p["k_length"]] * B) * (1.0 + 0.01 * np.arange(B))[:, None]
            args = (kind, ells, [p["k_scale"]] * B, [p["noise"]] * B, 1e-6, y)
            for _ in range(20):
                e.fit_batch(*args)
            t0 = time.perf_counter()
            for _ in range(200):
                e.fit_batch(*args)
            fb[f"B{B}_us"] = (time.perf_counter() - t0) / 200 * 1e6
        rec["fused" if mode == "1" else "general"] = {"host_factor_plus_grad_us": host_pair * 1e6, "device_classes_[launches,us]_per_step": dev,
                                                      "fit_batch_host": fb}
        e.close()
    f, g = rec["fused"], rec["general"]
    print(f"N={N:4d}: pair host {f['host_factor_plus_grad_us']:.0f} / {g['host_factor_plus_grad_us']:.0f} us  device fused kernel "
          f"{f['device_classes_[launches,us]_per_step']['potf2'][1]:.1f} us  fit_batch B1 {f['fit_batch_host']['B1_us']:.0f} / {g['fit_batch_host']['B1_us']:.0f}  "
          f"B64 {f['fit_batch_host']['B64_us']:.0f} / {g['fit_batch_host']['B64_us']:.0f}", file=sys.stderr, flush=True)
    out.append(rec)
print(json.dumps({"note": "fused / general; device time = HIP events around the fused kernel (counted under the potf2 class)", "cases": out}))
