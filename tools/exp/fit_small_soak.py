"""Soak of the fused small-N fit step (csrc/fit_small.hip): the same call repeated — alone, and from three contexts driven by three
host threads at once — must give the same bits every time (lml, gradient, alpha, pivot report), at every N where a branch of the
kernel changes."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.getcwd())
import bench_inputs
from gpax_amd import _lib

ok = True
engs = [_lib.Engine(0) for _ in range(3)]
for N, d, kind in [(5, 1, 0), (25, 1, 1), (31, 2, 0), (47, 3, 1), (63, 2, 2), (64, 1, 0), (100, 5, 1), (127, 2, 0)]:
    X, y, _, p = bench_inputs.synthetic_problem(N, d, 4, seed=N)
    ell = np.asarray(p["k_length"], dtype=float)
    if kind == 2:
        ell = np.concatenate([ell, [2.9]])
    B = 5
    ells = np.stack([ell * (1 + 0.03 * b) for b in range(B)])
    args = (kind, ells, [p["k_scale"]] * B, [p["noise"]] * B, 1e-6, y)
    for e in engs:
        e.set_train(X)
    ref = engs[0].fit_batch(*args)
    key = lambda r: (r[0].tobytes(), r[1].tobytes(), r[2].tobytes(), r[3].tobytes())
    k0 = key(ref)
    bad = 0
    for _ in range(300):
        bad += key(engs[0].fit_batch(*args)) != k0
    res = [0, 0, 0]

    def work(i):
        for _ in range(300):
            res[i] += key(engs[i].fit_batch(*args)) != k0

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    print(f"N={N} d={d} kind={kind}: alone {bad} / 300 differ, three contexts {res} / 300 differ", flush=True)
    ok = ok and bad == 0 and sum(res) == 0
print("fit_small_soak", "ok" if ok else "FAILED")
