"""Where a NUTS leapfrog's wall-clock goes at small N: device fit step vs host (Python) overhead."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import ExactGP, _lib
from gpax_amd.utils import get_keys
from oracle import cpu_ref as ref
import bench_inputs
for N in (128, 512, 2048):
    X, y, _, p = bench_inputs.synthetic_problem(N, 1, 4, seed=0)
    m = ExactGP(1, "Matern")
    t0 = time.perf_counter()
    m.fit(get_keys()[0], X, y, num_warmup=100, num_samples=100, progress_bar=False, print_summary=False)
    dt = time.perf_counter() - t0
    st = m.mcmc.get_extra_fields()[0]
    nl = int(np.sum(st["n_leapfrog"]))
    eng = _lib.get_engine()
    dev = eng.time_stage(_lib.STAGE_FITSTEP, 20) / 20
    print(f"N={N}: fit(100+100) {dt:.2f} s; sampling-phase leapfrogs {nl} (x2 for warm-up ~ {2*nl}); "
          f"~{dt/(2*nl)*1e3:.3f} ms per leapfrog vs device fit step {dev:.3f} ms", flush=True)
if "--profile" in sys.argv:
    X, y, _, p = bench_inputs.synthetic_problem(512, 1, 4, seed=0)
    m = ExactGP(1, "Matern")
    pr = cProfile.Profile(); pr.enable()
    m.fit(get_keys()[0], X, y, num_warmup=100, num_samples=100, progress_bar=False, print_summary=False)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
