"""Quick on-GPU timing probe (not a test): stage timings + per-class kernel profile."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpax_amd import _lib  # noqa: E402
from oracle import cpu_ref as ref  # noqa: E402  (synthetic inputs only)
import bench_inputs


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 16384]
    eng = _lib.Engine(0)
    out = {"device": eng.device_info()}
    out["mfma_f64_peak_tflops"] = eng.mfma_f64_peak()
    print(out, flush=True)
    # pure GEMM main-loop efficiency (device time via the profile events)
    rng = np.random.default_rng(0)
    for (Mg, Ng, Kg) in [(4096, 4096, 512), (4096, 4096, 4096), (8192, 8192, 512)]:
        A = rng.standard_normal((Mg, Kg)); B = rng.standard_normal((Ng, Kg))
        eng.gemm_nt(A, B)
        eng.profile_enable(True); eng.profile_reset()
        eng.gemm_nt(A, B)
        n, ms, work = eng.profile_read(1)
        eng.profile_enable(False)
        print(f"gemm {Mg}x{Ng}x{Kg}: {ms:.3f} ms  {2.0*Mg*Ng*Kg/(ms*1e-3)/1e12:.1f} TF", flush=True)
    for N in sizes:
        d, M = 2, 1024
        kind = 0 if N <= 4096 else 1
        X, y, Xnew, p = bench_inputs.synthetic_problem(N, d, M, seed=0)
        eng.set_train(X)
        t0 = time.time()
        lml, info = eng.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        t1 = time.time()
        mean, cov, var = eng.posterior(Xnew, p["noise"], 1e-6, want_cov=True, want_var=True)
        t2 = time.time()
        res = {"N": N, "lml": lml, "info": info, "factor_wall_s": t1 - t0, "posterior_wall_s": t2 - t1}
        for name, st in [("gram", _lib.STAGE_GRAM), ("potrf", _lib.STAGE_POTRF), ("posterior", _lib.STAGE_POSTERIOR),
                         ("predict", _lib.STAGE_PREDICT), ("fitstep", _lib.STAGE_FITSTEP)]:
            eng.time_stage(st, 1)
            reps = 3
            res[name + "_ms"] = eng.time_stage(st, reps) / reps
        Np = N
        res["potrf_tflops"] = (Np ** 3 / 3) / (res["potrf_ms"] * 1e-3) / 1e12
        res["fitstep_tflops"] = (Np ** 3) / (res["fitstep_ms"] * 1e-3) / 1e12
        post_flops = N ** 3 / 3 + N * N * M + N * M * M
        res["posterior_tflops"] = post_flops / (res["posterior_ms"] * 1e-3) / 1e12
        eng.profile_enable(True)
        eng.profile_reset()
        eng.time_stage(_lib.STAGE_POTRF, 1)
        prof = {}
        for cname, c in [("gemm_trailing", 0), ("gemm_other", 1), ("potf2", 2), ("gram", 3)]:
            n, ms, work = eng.profile_read(c)
            prof[cname] = {"launches": n, "ms": ms, "work": work,
                           "rate": (work / (ms * 1e-3) / 1e12) if ms > 0 else None}
        eng.profile_enable(False)
        res["potrf_profile"] = prof
        print(json.dumps(res), flush=True)
        out[f"N{N}"] = res
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/probe.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
