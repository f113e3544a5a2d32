"""The Gram build alone (gram_kernel<KIND, D>: the one HBM-bound kernel of the path, gpax/kernels/kernels.py:28-91) at the
bench size: HIP-event time of the stand-alone stage, bytes the lower 32 x 512 tiles actually store and the 8 N^2 the
symmetric build may claim, against the 8 TB/s HBM3E figure.  Run it plainly for the JSON line, or under
`rocprofv3 --kernel-trace --pmc WRITE_SIZE` / `FETCH_SIZE` for the counters (tools/profile_round.sh)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from bench import gram_bytes_written  # noqa: E402
from bench_inputs import synthetic_problem  # noqa: E402
from gpax_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kind = int(os.environ.get("GRAM_KIND", "1"))  # 1 = Matern: C3 (0 = RBF: one exp per entry, no square root)
X, y, Xn, p = synthetic_problem(N, 2, 8, seed=0)
e = _lib.Engine(0)
e.set_train(X)
e.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
e.time_stage(_lib.STAGE_GRAM, 2)
ms = [e.time_stage(_lib.STAGE_GRAM, 1) for _ in range(9)]
t = float(np.median(ms)) * 1e-3
Np = (N + 1 + 127) // 128 * 128
written = gram_bytes_written(N, Np)


def write_only_ceiling(nbytes):
    """What a pure write stream reaches on this device: hipMemsetAsync of the same number of bytes (the runtime's fill
    kernel), HIP events, best of 9 — the yardstick for a kernel that only WRITES (the 6.29 TB/s copy figure of
    MI355X_MICROARCH.md is read + write traffic together)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    ptr, e0, e1 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(ptr), C.c_size_t(nbytes)) == 0
    hip.hipEventCreate(C.byref(e0)); hip.hipEventCreate(C.byref(e1))
    best = 1e9
    for _ in range(12):
        hip.hipEventRecord(e0, None)
        hip.hipMemsetAsync(ptr, 0, C.c_size_t(nbytes), None)
        hip.hipEventRecord(e1, None)
        hip.hipEventSynchronize(e1)
        ms = C.c_float()
        hip.hipEventElapsedTime(C.byref(ms), e0, e1)
        best = min(best, ms.value)
    hip.hipFree(ptr)
    return nbytes / (best * 1e-3) / 1e9, best


fill_GBps, fill_ms = write_only_ceiling(written)
print(json.dumps({"kernel": "gpx::gram_kernel<1, 2> (Matern-5/2, d = 2), lower 32 x 512 tiles", "N": N, "median_ms": t * 1e3,
                  "runs_ms": ms, "bytes_written": written, "written_GBps": written / t / 1e9,
                  "frac_of_8TBps_on_written_bytes": written / t / 8e12, "alg_bytes_8N2": 8.0 * N * N,
                  "alg_GBps": 8.0 * N * N / t / 1e9, "bytes_read": 8.0 * 2 * N * 2,
                  "best_ms": float(min(ms)), "best_written_GBps": written / (min(ms) * 1e-3) / 1e9,
                  "write_only_fill_GBps": fill_GBps, "write_only_fill_ms": fill_ms,
                  "frac_of_write_only_fill": (written / t / 1e9) / fill_GBps,
                  "note": "the upper half is never written: 'alg' credits the full matrix to the symmetric build as SURVEY 8d "
                          "allows; the roofline figure to read is written_GBps"}))
