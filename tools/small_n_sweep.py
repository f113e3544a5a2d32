"""Experiment: predictive-sweep throughput at small N vs. the number of libgpx contexts in flight."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
from oracle import cpu_ref as ref
import bench_inputs
kind = 1
S = int(os.environ.get("S", "1024"))
ctxs = [int(c) for c in os.environ.get("CTX", "1,3,6,8,12").split(",")]
engs = [_lib.Engine(0) for _ in range(max(ctxs))]
SIZES = [(256, 1, 100), (512, 1, 100), (1024, 2, 256), (2048, 2, 1024), (4096, 2, 1024)]
if os.environ.get("SIZES"):  # e.g. SIZES=512,1,100
    SIZES = [tuple(int(v) for v in os.environ["SIZES"].split(","))]
for N, d, M in SIZES:
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=0)
    th = bench_inputs.synthetic_theta_samples(S, d, seed=1)
    eps = np.random.default_rng(2).standard_normal((S, 1, M))
    for n in ctxs:
        _lib.concurrent_sweep(engs[:n], X, kind, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)  # warm (allocations)
        t0 = time.perf_counter()
        _lib.concurrent_sweep(engs[:n], X, kind, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)
        dt = time.perf_counter() - t0
        print(f"N={N} d={d} M={M}: {n:2d} ctx: {S/dt:8.1f} posteriors/s  ({dt/S*1e3:.3f} ms each)", flush=True)
