"""The K = 128 lower-tile update of a chain step (C -= P P^T, t x t tile rows) on the 64 x 64 latency shape against the
128 x 128 throughput shape, t = 1 .. 40: ms per launch (back-to-back launches on resident operands, gpx_debug_gemm_time)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from gpax_amd import _lib  # noqa: E402

e = _lib.Engine(0)
out = {}
for t in list(range(1, 33)) + [36, 40]:
    lat = min(e.gemm_time(t, t, 128, 1, True, 1, 50) for _ in range(3))
    big = min(e.gemm_time(t, t, 128, 1, True, 2, 50) for _ in range(3))
    out[t] = {"lat_us": round(1e3 * lat, 2), "big_us": round(1e3 * big, 2)}
    print(t, out[t], file=sys.stderr, flush=True)
e.close()
print(json.dumps(out))
