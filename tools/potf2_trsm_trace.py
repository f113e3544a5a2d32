"""Where the time of a fused chain step (potf2.hip potf2_trsm_kernel) goes.  Needs the trace build of the library:
    make -C gpax_amd/csrc trace && GPX_LIB=gpax_amd/lib/libgpx_trace.so python tools/potf2_trsm_trace.py [N ...]
Every launch leaves 100 MHz wall-clock stamps: workgroup 0's start / end of factorisation / flag published, the first and
last strip workgroup seeing the flag, the last strip with L^-1 in LDS, the last strip stored.  Printed per N (one
factorisation, after warm-up): medians over the steps, and per group of steps by the number of strips."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from gpax_amd import _lib  # noqa: E402

RING = 1024
lib = _lib.load_library()
eng = _lib.Engine(0)
rng = np.random.default_rng(0)
out = {}
for N in [int(v) for v in sys.argv[1:]] or [512, 2048, 4096, 5120]:
    B = rng.standard_normal((N, N + 8))
    A = B @ B.T / N + 0.3 * np.eye(N)
    for _ in range(3):
        eng.potrf(A)
    buf = np.zeros(RING * 8, dtype=np.int64)
    n = lib.gpx_debug_chain_trace(buf.ctypes.data_as(C.POINTER(C.c_longlong)), RING)
    assert n == RING
    r = buf.reshape(RING, 8).astype(np.float64)
    r = r[(r[:, 7] > 0) & (r[:, 6] > 0)]
    steps = N // 128 - 1
    # the last `steps` launches by start time
    r = r[np.argsort(r[:, 0])][-steps:]
    us = lambda a: a / 100.0
    rec = {
        "steps": int(len(r)),
        "potf2_us": float(np.median(us(r[:, 1] - r[:, 0]))),
        "publish_us": float(np.median(us(r[:, 2] - r[:, 1]))),
        "flag_to_first_strip_us": float(np.median(us(r[:, 3] - r[:, 2]))),
        "flag_to_last_strip_us": float(np.median(us(r[:, 4] - r[:, 2]))),
        "last_strip_linv_in_lds_us": float(np.median(us(r[:, 5] - r[:, 4]))),
        "last_strip_mfma_store_us": float(np.median(us(r[:, 6] - r[:, 5]))),
        "after_potf2_total_us": float(np.median(us(r[:, 6] - r[:, 1]))),
        "step_us": float(np.median(us(r[:, 6] - r[:, 0]))),
    }
    groups = {}
    for lo, hi in ((1, 64), (65, 128), (129, 192), (193, 256), (257, 10**6)):
        g = r[(r[:, 7] >= lo) & (r[:, 7] <= hi)]
        if len(g):
            groups[f"{lo}-{min(hi, int(r[:, 7].max()))} strips"] = {
                "n": int(len(g)), "potf2_us": float(np.median(us(g[:, 1] - g[:, 0]))),
                "after_potf2_total_us": float(np.median(us(g[:, 6] - g[:, 1]))),
                "flag_to_last_strip_us": float(np.median(us(g[:, 4] - g[:, 2]))),
                "linv_in_lds_us": float(np.median(us(g[:, 5] - g[:, 4]))),
                "mfma_store_us": float(np.median(us(g[:, 6] - g[:, 5])))}
    rec["by_strips"] = groups
    out[N] = rec
    print(N, json.dumps(rec, indent=1), flush=True)
