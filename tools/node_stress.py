"""Host-side stress of the two multi-GPU launch models on a 1-GPU box at FULL C4 size ("threads x 3 contexts x
8 'devices' — host-side races show up here"): the node-level sweep (one process) over the memcpy test transport with GPU 0
listed G times, against the single-GPU sweep; every block must be bit-identical.  G from argv (default 8), S from env."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["GPX_NODE_TRANSPORT"] = "memcpy"
import bench_inputs
from gpax_amd import _lib
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(os.environ.get("S", "1000"))
N, d, M = 8192, 3, 1024
X, y, Xn, _ = bench_inputs.synthetic_problem(N, d, M, seed=0)
th = bench_inputs.synthetic_theta_samples(S, d, seed=1)
eps = np.random.default_rng(2).standard_normal((S, 1, M))
node = _lib.Node([0] * G, inflight=3)
node.predict_sweep(X, 1, th["k_length"][:G * 24], th["k_scale"][:G * 24], th["noise"][:G * 24], y, Xn, False, 1e-6, eps[:G * 24])
t0 = time.perf_counter()
got = node.predict_sweep(X, 1, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)
dt = time.perf_counter() - t0
node.close()
engines = [_lib.Engine(0) for _ in range(3)]
want = _lib.concurrent_sweep(engines, X, 1, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)
same = all(np.array_equal(a, b) for a, b in zip(got, want))
print(json.dumps({"test": "node memcpy transport, GPU 0 listed G times, full C4", "G": G, "contexts": 3 * G, "S": S, "seconds": dt,
                  "posteriors_per_s": S / dt, "identical_to_one_gpu": bool(same), "nan_rows": int(np.isnan(got[1]).any(axis=(1, 2)).sum())}))
assert same
