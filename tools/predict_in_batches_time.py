"""ExactGP.predict_in_batches: slices as covariance blocks of one sweep (one factorisation per sample) vs the
reference's loop (one predict per slice, gp.py:325-349)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import ExactGP
from gpax_amd.utils import get_keys
from oracle import cpu_ref as ref
import bench_inputs
for N, M, bs, S in [(1024, 10000, 1000, 200), (4096, 8000, 1000, 60), (8192, 8000, 1000, 30)]:
    X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, M, seed=0)
    th = bench_inputs.synthetic_theta_samples(S, 2, seed=1)
    m = ExactGP(2, "Matern")
    m.X_train, m.y_train = m._set_data(X, y)
    key = get_keys()[1]
    for rep in range(2):
        t0 = time.perf_counter(); a = m.predict_in_batches(key, Xn, batch_size=bs, samples=th, n=1); t1 = time.perf_counter()
        b = m.predict_in_batches(key, Xn, batch_size=bs, samples=th, n=1, predict_fn=lambda xi: m.predict(key, xi, th, n=1))
        t2 = time.perf_counter()
    same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    print(f"N={N} M={M} batch_size={bs} S={S}: one sweep {t1 - t0:.3f} s, slice-by-slice {t2 - t1:.3f} s ({(t2 - t1) / (t1 - t0):.1f}x), identical={same}", flush=True)
