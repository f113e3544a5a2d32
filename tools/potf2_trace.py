"""Where the diagonal-block kernel's time goes INSIDE the pipeline (round 4).  Needs the trace build of the library:
    make -C gpax_amd/csrc trace && GPX_LIB=gpax_amd/lib/libgpx_trace.so python tools/potf2_trace.py [N]
Every potf2_slim launch leaves 100 MHz wall-clock stamps of all four waves at their barriers (potf2_slim.h,
GPX_POTF2_TRACE).  Printed per group of launches (stand-alone: potrf of a 128 x 128 matrix; in the pipeline: the
launches of C3-sized factorisations): time inside the kernel, the chain wave's diag16 / T+U phases, how long it waited
at the two barriers for the workers, and the workers' solve / update windows — the split that tells contention on the
chain wave from waiting for the memory-resident tiles of the workers."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from gpax_amd import _lib  # noqa: E402
from bench_inputs import synthetic_problem  # noqa: E402

RING, STAMPS = 512, 40


def read_trace(lib):
    buf = np.zeros(RING * 4 * STAMPS, dtype=np.int64)
    cnt = (C.c_uint * 4)()
    n = lib.gpx_debug_slim_trace(buf.ctypes.data_as(C.POINTER(C.c_longlong)), RING, cnt)
    assert n >= 0
    return n, buf.reshape(RING, 4, STAMPS)


def summarise(tag, recs):
    recs = np.asarray(recs, dtype=np.float64) / 100.0  # us
    ch, w = recs[:, 0, :], recs[:, 1:, :]
    out = {"group": tag, "launches": int(len(recs))}
    out["in_kernel_us"] = float(np.median(ch[:, 33] - ch[:, 0]))
    out["in_kernel_us_mean"] = float(np.mean(ch[:, 33] - ch[:, 0]))
    out["wave_start_skew_us"] = float(np.median(np.max(recs[:, :, 0], axis=1) - np.min(recs[:, :, 0], axis=1)))
    d16, wb1, tu, wb2 = [], [], [], []
    for P in range(8):
        prev = ch[:, 4 * P] if P > 0 else ch[:, 0]
        d16.append(ch[:, 1 + 4 * P] - prev)        # start tiles (P = 0) + diag16(P)
        wb1.append(ch[:, 2 + 4 * P] - ch[:, 1 + 4 * P])  # chain waits at B1 for the workers' update window
        tu.append(ch[:, 3 + 4 * P] - ch[:, 2 + 4 * P])
        wb2.append(ch[:, 4 + 4 * P] - ch[:, 3 + 4 * P])  # chain waits at B2 for the workers' solve
    out["chain_diag16_us_per_panel"] = [round(float(np.median(x)), 2) for x in d16]
    out["chain_wait_B1_us_per_panel"] = [round(float(np.median(x)), 2) for x in wb1]
    out["chain_TU_us_per_panel"] = [round(float(np.median(x)), 2) for x in tu]
    out["chain_wait_B2_us_per_panel"] = [round(float(np.median(x)), 2) for x in wb2]
    out["chain_sum_us"] = {k: round(float(np.median(sum(v))), 2) for k, v in
                           (("diag16", d16), ("wait_B1", wb1), ("TU", tu), ("wait_B2", wb2))}
    # workers: update window = from leaving B2(P) to arriving at B1(P+1); solve = leaving B1(P) to arriving at B2(P)
    upd = [np.max(w[:, :, 1 + 4 * (P + 1)] - w[:, :, 4 + 4 * P], axis=1) for P in range(7)]
    sol = [np.max(w[:, :, 3 + 4 * P] - w[:, :, 2 + 4 * P], axis=1) for P in range(8)]
    out["worker_update_window_us_per_panel"] = [round(float(np.median(x)), 2) for x in upd]
    out["worker_solve_us_per_panel"] = [round(float(np.median(x)), 2) for x in sol]
    out["worker_prologue_us"] = round(float(np.median(np.max(w[:, :, 1] - w[:, :, 0], axis=1))), 2)
    return out


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    lib = _lib.load_library()
    lib.gpx_debug_slim_trace.restype = C.c_int
    lib.gpx_debug_slim_trace.argtypes = [C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_uint)]
    eng = _lib.Engine(0)
    rng = np.random.default_rng(0)
    A = rng.standard_normal((128, 128))
    A = A @ A.T + 128 * np.eye(128)
    for _ in range(40):
        eng.potrf(A)
    n0, tr = read_trace(lib)
    alone = [tr[i % RING] for i in range(n0 - 30, n0)]
    res = [summarise("stand-alone (potrf of one 128 x 128 block)", alone)]
    X, y, Xn, p = synthetic_problem(N, 2, 1024, seed=0)
    eng.set_train(X)
    for rep in range(3):
        eng.factor(1, p["k_length"], p["k_scale"], p["noise"] * (1 + 1e-3 * rep), 1e-6, y)
    n1, tr = read_trace(lib)
    per = (n1 - n0) // 3
    last = [tr[i % RING] for i in range(n1 - per, n1)]  # the launches of the last factorisation, in launch order
    res.append(summarise(f"N = {N} factorisation, all {per} diagonal blocks", last))
    res.append(summarise("... its first half (GEMM-bound head: a trailing update always resident)", last[:per // 2]))
    res.append(summarise("... its last eighth (chain-bound tail)", last[-per // 8:]))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
