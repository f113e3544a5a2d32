"""Experiment: host-side cost of one fit step through the C-ABI at small / mid N (wall per call vs device time)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
import bench_inputs
eng = _lib.Engine(0)
for N in [int(v) for v in os.environ.get("NS", "128,512,1024,2048,4096").split(",")]:
    X, y, _, p = bench_inputs.synthetic_problem(N, 1, 4, seed=0)
    eng.set_train(X)
    ells = np.array([[p["k_length"][0]]]); sc = np.array([p["k_scale"]]); nz = np.array([p["noise"]])
    for _ in range(5):
        eng.fit_batch(1, ells, sc, nz, 1e-6, y)
    R = int(os.environ.get("R", "200"))
    t0 = time.perf_counter()
    for i in range(R):
        eng.fit_batch(1, ells * (1 + 1e-9 * i), sc, nz, 1e-6, y)
    wall = (time.perf_counter() - t0) / R * 1e3
    dev = eng.time_stage(_lib.STAGE_FITSTEP, 20) / 20
    print(f"N={N}: fit_batch wall {wall:.3f} ms per call, device fit step (queued back to back) {dev:.3f} ms", flush=True)
