"""Margin of tests/test_gpu_hskgp.py::test_varnoise_gp_on_gpu_recovers_heteroskedastic_noise on the GPU: the ratio the
test thresholds at 1.5, for the test's own key and two more."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import VarNoiseGP
from gpax_amd.utils import get_keys
rng = np.random.default_rng(0)
N = 60
X = np.sort(rng.uniform(0, 6, N))
sd = 0.05 + 0.12 * X
y = np.sin(1.5 * X) + sd * rng.standard_normal(N)
for seed in (0, 1, 2):
    k1, k2 = get_keys(seed)
    m = VarNoiseGP(1, "RBF", noise_kernel="RBF")
    t = time.time()
    m.fit(k1, X, y, num_warmup=200, num_samples=60, progress_bar=False, print_summary=False)
    v = np.median(m.get_data_var_samples(), axis=0)
    print("seed", seed, "ratio %.2f" % (np.mean(v[X > 4]) / np.mean(v[X < 2])), "%.1f s" % (time.time() - t), flush=True)
