"""Stand-alone duration of the 128x128 diagonal-block kernel (HIP events around the launch, GPX_POTF2 variant)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
eng = _lib.Engine(0)
rng = np.random.default_rng(0)
for n in (128, 256, 2048):
    A = rng.standard_normal((n, n)); A = A @ A.T + n * np.eye(n)
    for _ in range(3):
        eng.potrf(A)
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(20):
        eng.potrf(A)
    nl, ms, _ = eng.profile_read(_lib.PROF_POTF2)
    eng.profile_enable(False)
    print(f"{os.environ.get('GPX_POTF2', 'default')}: n={n}: {nl} launches, {ms / nl * 1e3:.1f} us per potf2 launch", flush=True)
