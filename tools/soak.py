"""Soak: many sweeps / fit steps / model calls in one process; per-call time and device memory must stay flat."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import ExactGP, _lib
from gpax_amd.utils import get_keys
from oracle import cpu_ref as ref
import bench_inputs


def vram_used():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmeminfo", "vram"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if "Used" in line:
            return int(line.split(":")[-1].strip()) / 2 ** 20
    return float("nan")


eng = _lib.get_engine()
rng = np.random.default_rng(0)
sizes = [(300, 2, 64), (700, 1, 130), (512, 3, 100), (1500, 2, 256)]
t_first, t_last = {}, {}
print(f"start: VRAM used {vram_used():.0f} MiB", flush=True)
for rnd in range(int(os.environ.get("ROUNDS", "40"))):
    for (N, d, M) in sizes:
        X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=N)
        th = bench_inputs.synthetic_theta_samples(64, d, seed=1)
        eps = rng.standard_normal((64, 1, M))
        t0 = time.perf_counter()
        eng.set_train(X)
        eng.predict_sweep(1, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps, m_slice=50 if rnd % 2 else 0)
        eng.fit_batch(1, th["k_length"][:8], th["k_scale"][:8], th["noise"][:8], 1e-6, y)
        eng.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        eng.lml_grad()
        eng.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        eng.posterior(Xn, p["noise"], 1e-6, want_cov=True)
        dt = time.perf_counter() - t0
        (t_first if rnd == 1 else t_last)[(N, d, M)] = dt
    if rnd in (1, 10, 20, 39):
        print(f"round {rnd}: VRAM used {vram_used():.0f} MiB", flush=True)
for k in sizes:
    print(f"{k}: round 1 {t_first[k]*1e3:.2f} ms, last round {t_last[k]*1e3:.2f} ms")
m = ExactGP(1, "RBF")
Xs, ys = np.linspace(0, 5, 80), np.sin(np.linspace(0, 5, 80))
for i in range(5):
    m.fit(get_keys(i)[0], Xs, ys, num_warmup=30, num_samples=30, progress_bar=False, print_summary=False)
    m.predict(get_keys(i)[1], np.linspace(0, 5, 40), n=2)
print(f"after 5 fits: VRAM used {vram_used():.0f} MiB")
