"""C4 (BASELINE.json configs[3]): the 1000-sample posterior predictive sweep at N=8192, d=3, M=1024, n=1 on ONE GPU
through the public ExactGP.predict (PCIe in/out, host reshuffles included).  SURVEY.md 8(d): 2.61e11 flop / posterior."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import ExactGP, _lib
from gpax_amd.utils import get_keys
import bench_inputs  # BASELINE.md 3 workloads
S = int(os.environ.get("S", "1000"))
N, d, M = 8192, 3, 1024
X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=0)
th = bench_inputs.synthetic_theta_samples(S, d, seed=1)
m = ExactGP(d, "Matern")
m.X_train, m.y_train = m._set_data(X, y)
samples = {"k_length": th["k_length"], "k_scale": th["k_scale"], "noise": th["noise"]}
k1, k2 = get_keys()
m.predict(k2, Xn, {k: v[:32] for k, v in samples.items()}, n=1)  # warm-up: allocations, clocks
t0 = time.perf_counter()
ym, ys = m.predict(k2, Xn, samples, n=1)
dt = time.perf_counter() - t0
flop = 2.61e11
rec = dict(config="C4", N=N, d=d, M=M, S=S, seconds=dt, posteriors_per_s=S / dt, tflops=S * flop / dt / 1e12,
           frac_of_fp64_peak=S * flop / dt / 78.6e12, batch=_lib.get_engine().sweep_stats()[2],
           contexts=len(_lib.get_sweep_engines()), nan_rows=int(np.isnan(ys).any(axis=(1, 2)).sum()))
print(json.dumps(rec))  # the caller redirects this line into the record (profiles/<round>/c4_sweep.json)
