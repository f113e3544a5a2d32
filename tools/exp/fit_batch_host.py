"""gpx_fit_batch (B = 1) host to host against the device time of the fit step, N = 200 ... 4096 (general launch sequence)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
import bench_inputs
eng = _lib.Engine(0)
for N, d in [(200, 1), (512, 1), (1024, 2), (2048, 2), (4096, 2)]:
    X, y, _, p = bench_inputs.synthetic_problem(N, d, 4, seed=1)
    eng.set_train(X)
    args = (0, np.asarray(p["k_length"], dtype=float)[None, :], [p["k_scale"]], [p["noise"]], 1e-6, y)
    for _ in range(20):
        eng.fit_batch(*args)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(30):
            eng.fit_batch(*args)
        ts.append((time.perf_counter() - t0) / 30)
    eng.factor(0, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    eng.lml_grad()
    eng.time_stage(_lib.STAGE_FITSTEP, 3)
    dev = eng.time_stage(_lib.STAGE_FITSTEP, 20) / 20
    print(f"N={N}: fit_batch host to host {np.median(ts)*1e3:.3f} ms, device fit step {dev:.3f} ms", flush=True)
