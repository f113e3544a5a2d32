"""Soak of the sparse step's two-stream forward pass (K_fu on the panel stream beside the K_uu chain): two alternating
hyper-parameter vectors, many repetitions — every repetition of one vector must give the same bits (bound, gradients)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from gpax_amd import _lib
rng = np.random.default_rng(0)
ok = True
for N, M in ((3000, 300), (16316, 2039), (9000, 1100)):
    X = rng.uniform(0, 100, (N, 2)); y = np.sin(X[:, 0] / 9) + 0.1 * rng.standard_normal(N)
    Xu = X[rng.choice(N, M, replace=False)].copy()
    eng = _lib.Engine(0); eng.set_train(X)
    ref = {}
    for rep in range(10):
        for k, noise in enumerate((0.05, 0.0501)):
            bound, info, g = eng.sgp_bound(1, [12.0, 12.0], 1.0, noise, 1e-6, Xu, y, True)
            key = (bound, info) + tuple(np.asarray(g[n], dtype=np.float64).tobytes() for n in sorted(g))
            if k not in ref:
                ref[k] = key
            elif key != ref[k]:
                ok = False
                print(f"N={N} M={M}: repetition {rep} of vector {k} differs", flush=True)
    print(f"N={N} M={M}: 10 x 2 alternating steps identical: {ok}", flush=True)
    eng.close()
print("sgp_soak", "ok" if ok else "FAILED")
