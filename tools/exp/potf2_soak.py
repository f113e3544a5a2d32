"""Soak of the wave-specialised diagonal-block kernel (potf2_chain.h) against the four-phase kernel it replaces: many
factorisations of random SPD matrices, bit-compared, while another context keeps the chip busy with trailing-update-sized
GEMM work (the contended situation inside a real factorisation).  Prints one JSON line."""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench_inputs
from gpax_amd import _lib
os.environ["GPX_POTF2"] = "tile"
e_tile = _lib.Engine(0)
os.environ["GPX_POTF2"] = "chain"
e_chain = _lib.Engine(0)
os.environ.pop("GPX_POTF2")
load = _lib.Engine(0)
X, y, Xn, p = bench_inputs.synthetic_problem(6144, 2, 256, seed=0)
load.set_train(X)
stop = threading.Event()
def busy():
    while not stop.is_set():
        load.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
t = threading.Thread(target=busy); t.start()
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
iters = int(os.environ.get("ITERS", "400"))
bad = 0; n_calls = 0; t0 = time.perf_counter()
for it in range(iters):
    n = int(rng.choice([40, 128, 129, 300, 512, 1000]))
    B = rng.standard_normal((n, n + 4))
    A = B @ B.T / n + (10.0 ** rng.uniform(-6, 0)) * np.eye(n)   # conditioning from 1e0 to ~1e7
    if it % 50 == 49:
        A[n // 2, n // 2] = -1.0   # a failing pivot now and then
    L0, i0 = e_tile.potrf(A)
    L1, i1 = e_chain.potrf(A)
    n_calls += 1
    if i0 != i1 or not np.array_equal(L0, L1, equal_nan=True):
        bad += 1
stop.set(); t.join()
print(json.dumps({"soak": "potf2 chain vs tile kernel under GEMM load", "factorisations": n_calls, "mismatches": bad,
                  "seconds": time.perf_counter() - t0}))
assert bad == 0
