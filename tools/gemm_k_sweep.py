"""Stand-alone rate of the 128x128 NT GEMM kernel vs the reduction length K (random data, beta = 0)."""
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
eng = _lib.Engine(0)
rng = np.random.default_rng(0)
M = N = 16384 if len(sys.argv) < 2 else int(sys.argv[1])
for K in (128, 256, 512, 1024, 2048):
    A, B = rng.standard_normal((M, K)), rng.standard_normal((N, K))
    for beta in (0.0, 1.0):
        C = rng.standard_normal((M, N)) if beta else np.zeros((M, N))
        eng.gemm_nt(A, B, 1.0, beta, C.copy()) if beta else eng.gemm_nt(A, B)
        best = 1e9
        for r in range(2):
            eng.profile_enable(True); eng.profile_reset()
            eng.gemm_nt(A, B, 1.0, beta, C.copy()) if beta else eng.gemm_nt(A, B)
            n, ms, work = eng.profile_read(1)
            eng.profile_enable(False)
            best = min(best, ms)
        print(f"K={K:5d} beta={beta}: {best:8.3f} ms  {2.0 * M * N * K / (best * 1e-3) / 1e12:5.1f} TF", flush=True)
