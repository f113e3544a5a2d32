"""Experiment: n libgpx contexts on one GPU sweeping disjoint theta shards concurrently."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
from oracle import cpu_ref as ref
import bench_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2
M = 1024
kind = 1
X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=0)
K = 24
th = bench_inputs.synthetic_theta_samples(K + 4, d, seed=1)
def setup():
    e = _lib.Engine(0); e.set_train(X); e.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    e.posterior(Xn, p["noise"], 1e-6, want_cov=True); e.mvn_draw(np.zeros((1, M)))
    e.sweep_resident(kind, th["k_length"][:2], th["k_scale"][:2], th["noise"][:2], False, 1e-6, 1)
    return e
engs = [setup() for _ in range(4)]
for n in [1, 2, 3, 4]:
    per = K // n
    def run(e, sl):
        e.sweep_resident(kind, th["k_length"][sl], th["k_scale"][sl], th["noise"][sl], False, 1e-6, 1)
    ts = [threading.Thread(target=run, args=(engs[i], slice(4 + i * per, 4 + (i + 1) * per))) for i in range(n)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print(f"N={N} d={d}: {n} ctx: {per*n} steps {dt*1e3:.1f} ms -> {per*n/dt:.2f}/s", flush=True)
