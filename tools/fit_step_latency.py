"""Experiment: latency of one NUTS leapfrog's device work (lml + gradient) vs N."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
from oracle import cpu_ref as ref
import bench_inputs
eng = _lib.Engine(0)
for N, d in [(128, 1), (256, 1), (512, 1), (1024, 2), (2048, 2), (4096, 2)]:
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, 16, seed=0)
    eng.set_train(X)
    eng.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    eng.lml_grad()
    eng.time_stage(_lib.STAGE_FITSTEP, 2)
    dev = eng.time_stage(_lib.STAGE_FITSTEP, 20) / 20
    pot = eng.time_stage(_lib.STAGE_POTRF, 20) / 20
    t0 = time.perf_counter()
    for _ in range(50):
        eng.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        eng.lml_grad()
    host = (time.perf_counter() - t0) / 50 * 1e3
    print(f"N={N}: potrf {pot:.3f} ms, fit step device {dev:.3f} ms, host-inclusive factor+lml_grad {host:.3f} ms", flush=True)
