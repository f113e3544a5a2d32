"""Device time of the C3 / N = 8192 fit step and predict (gpx_time_stage), median of 5."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
import bench_inputs
eng = _lib.Engine(0)
for N, d, kind in [(8192, 3, 1), (16384, 2, 1)]:
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, 1024, seed=0)
    eng.set_train(X)
    eng.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    eng.lml_grad()
    eng.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    eng.posterior(Xn, p["noise"], 1e-6, want_cov=True)
    eng.mvn_draw(np.zeros((1, 1024)))
    for name, st in (("fit_step", _lib.STAGE_FITSTEP), ("predict", _lib.STAGE_PREDICT)):
        eng.time_stage(st, 1)
        print(N, name, "%.3f ms" % np.median([eng.time_stage(st, 1) for _ in range(5)]), flush=True)
