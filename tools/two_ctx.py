"""Experiment: two libgpx contexts on one GPU sweeping disjoint theta shards with a phase offset."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
from oracle import cpu_ref as ref
import bench_inputs
N, d, M = 16384, 2, 1024
X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=0)
K = 12
th = bench_inputs.synthetic_theta_samples(2 * K + 4, d, seed=1)
def setup():
    e = _lib.Engine(0); e.set_train(X); e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    e.posterior(Xn, p["noise"], 1e-6, want_cov=True); e.mvn_draw(np.zeros((1, M)))
    e.sweep_resident(1, th["k_length"][:2], th["k_scale"][:2], th["noise"][:2], False, 1e-6, 1)
    return e
e1 = setup()
t0 = time.perf_counter(); e1.sweep_resident(1, th["k_length"][4:4+2*K], th["k_scale"][4:4+2*K], th["noise"][4:4+2*K], False, 1e-6, 1); t1 = time.perf_counter() - t0
print(f"one ctx: {2*K} steps {t1*1e3:.1f} ms -> {2*K/t1:.2f}/s", flush=True)
e2 = setup()
for off_ms in [0, 10, 20, 30]:
    def run(e, sl, delay):
        time.sleep(delay)
        e.sweep_resident(1, th["k_length"][sl], th["k_scale"][sl], th["noise"][sl], False, 1e-6, 1)
    ta = threading.Thread(target=run, args=(e1, slice(4, 4 + K), 0.0))
    tb = threading.Thread(target=run, args=(e2, slice(4 + K, 4 + 2 * K), off_ms * 1e-3))
    t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); dt = time.perf_counter() - t0
    print(f"two ctx offset {off_ms} ms: {2*K} steps {dt*1e3:.1f} ms -> {2*K/dt:.2f}/s", flush=True)
