"""Run ONE factorisation (Gram + potrf) at N under rocprofv3 --kernel-trace and dump the kernel timeline."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
import bench_inputs
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
eng = _lib.Engine(0)
X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, 1024, seed=0)
eng.set_train(X)
eng.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)   # warm-up
eng.synchronize()
print("MARK", flush=True)
eng.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
eng.synchronize()
