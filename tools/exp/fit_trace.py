"""Experiment: 3 fit steps (factor + lml_grad) at C3 size under rocprofv3 --kernel-trace; the per-launch list of the last one."""
import os, sys
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
import bench_inputs
N = int(os.environ.get("N", "16384"))
eng = _lib.Engine(0)
X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, 16, seed=0)
eng.set_train(X)
for _ in range(3):
    eng.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    eng.lml_grad()
