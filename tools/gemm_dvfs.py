import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
eng = _lib.Engine(0)
rng = np.random.default_rng(0)
M = N = 8192; K = 2048
for name, A, B in [("random", rng.standard_normal((M, K)), rng.standard_normal((N, K))),
                   ("zeros", np.zeros((M, K)), np.zeros((N, K))),
                   ("ones", np.ones((M, K)), np.ones((N, K))),
                   ("random_small_int", rng.integers(-2, 3, (M, K)).astype(float), rng.integers(-2, 3, (N, K)).astype(float))]:
    eng.gemm_nt(A, B)
    best = 1e9
    for r in range(3):
        eng.profile_enable(True); eng.profile_reset()
        eng.gemm_nt(A, B)
        n, ms, work = eng.profile_read(1)
        eng.profile_enable(False)
        best = min(best, ms)
    print(f"{name}: {best:.3f} ms {2.0*M*N*K/(best*1e-3)/1e12:.1f} TF", flush=True)
