"""Soak of the round-4 schedule changes (one outer block up to 40 tile rows; the end of a larger factorisation finished as
one block behind an event wait on the main stream): the same theta factored again and again, alone and with three
contexts in flight, must give the same bits every time — a missing cross-stream dependency would show as a changed
log-likelihood / posterior mean on some repetition."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from gpax_amd import _lib
import bench_inputs

ok = True
for N, M, reps in ((5120, 256, 12), (6144, 300, 12), (8192, 1024, 10), (16384, 1024, 6), (16384, 2048, 4), (11000, 700, 6)):
    X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, M, seed=N)
    engines = [_lib.Engine(0) for _ in range(3)]
    for e in engines:
        e.set_train(X)
    ref = None
    # alone
    for r in range(reps):
        lml, info = engines[0].factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        mean, cov, _ = engines[0].posterior(Xn, p["noise"], 1e-6)
        key = (lml, mean.tobytes(), np.diag(cov).tobytes())
        if ref is None:
            ref = key
        elif key != ref:
            ok = False
            print(f"N={N} M={M}: repetition {r} alone differs (lml {lml!r} vs {ref[0]!r})", flush=True)
    # three contexts in flight
    out = [[] for _ in engines]
    def work(i):
        for r in range(reps):
            lml, info = engines[i].factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
            mean, cov, _ = engines[i].posterior(Xn, p["noise"], 1e-6)
            out[i].append((lml, mean.tobytes(), np.diag(cov).tobytes()))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    bad = sum(1 for o in out for k in o if k != ref)
    if bad:
        ok = False
    print(f"N={N} M={M}: {reps} repetitions alone + 3 x {reps} in flight: {'identical' if not bad else str(bad) + ' DIFFER'}; lml {ref[0]!r}", flush=True)
    for e in engines:
        e.close()
print("finish_soak", "ok" if ok else "FAILED")
