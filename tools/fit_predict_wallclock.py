"""End-to-end fit + predict wall-clock through the public model API (BASELINE.json metric, first half):
ExactGP (NUTS) and viGP (SVI) at C1 (RBF, N=512, d=1: the reference's own CPU-sized case, 200 + 200 NUTS draws), C2 (RBF,
N=4096) and C3 (Matern, N=16384), d=2, M=1024, synthetic data.  Each config in a process of its own (argv names the
configs): the first call at a size pays that size's device allocations."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import ExactGP, viGP
from gpax_amd.utils import get_keys
import bench_inputs  # BASELINE.md 3 workloads
out = []
for name, kernel, N, d, nuts, svi in [("C1", "RBF", 512, 1, (200, 200), 200), ("C2", "RBF", 4096, 2, (20, 20), 100),
                                     ("C3", "Matern", 16384, 2, (5, 5), 30)]:
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, 1024, seed=0)
    k1, k2 = get_keys()
    m = ExactGP(d, kernel)
    t0 = time.perf_counter()
    m.fit(k1, X, y, num_warmup=nuts[0], num_samples=nuts[1], progress_bar=False, print_summary=False)
    t1 = time.perf_counter()
    ym, ys = m.predict(k2, Xn, n=1)
    t2 = time.perf_counter()
    nl = int(sum(int(np.sum(st["n_leapfrog"])) for st in m.mcmc.get_extra_fields()))  # sampling phase only
    rmse = float(np.sqrt(np.mean((ym - np.prod(np.sin(Xn + 0.3 * np.arange(d)), axis=1)) ** 2)))
    rec = dict(config=name, model="ExactGP", kernel=kernel, N=N, d=d, M=1024, num_warmup=nuts[0], num_samples=nuts[1],
               fit_ms=(t1 - t0) * 1e3, predict_ms=(t2 - t1) * 1e3, posteriors=nuts[1],
               leapfrogs_in_sampling=nl, rmse_vs_truth=rmse)
    print(json.dumps(rec), flush=True)
    out.append(rec)
    v = viGP(d, kernel)
    t0 = time.perf_counter()
    v.fit(k1, X, y, num_steps=svi, step_size=2e-2, progress_bar=False, print_summary=False)
    t1 = time.perf_counter()
    mean, var = v.predict(k2, Xn)
    t2 = time.perf_counter()
    rmse = float(np.sqrt(np.mean((mean - np.prod(np.sin(Xn + 0.3 * np.arange(d)), axis=1)) ** 2)))
    rec = dict(config=name, model="viGP", kernel=kernel, N=N, M=1024, num_steps=svi, fit_ms=(t1 - t0) * 1e3,
               ms_per_svi_step=(t1 - t0) * 1e3 / svi, predict_ms=(t2 - t1) * 1e3, rmse_vs_truth=rmse)
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs("gpurun_out/prof", exist_ok=True)
path = "gpurun_out/prof/fit_predict_wallclock.json"
if len(sys.argv) > 1 and os.path.exists(path):  # a run restricted to some configs keeps the others' records
    done = {r["config"] for r in out}
    out = [r for r in json.load(open(path)) if r["config"] not in done] + out
json.dump(out, open(path, "w"), indent=1)
