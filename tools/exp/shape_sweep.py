"""64 x 64 latency shape against the 128 x 128 throughput shape over (tile rows x tile columns, K), full and lower grids,
beta = 0 and the update form: ms per launch on resident operands (gpx_debug_gemm_time), alone on the chip."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from gpax_amd import _lib  # noqa: E402

e = _lib.Engine(0)
out = []
for lower in (False, True):
    for mode in (0, 1):
        for K in (128, 256, 512, 1024, 2048, 4096):
            for t in (8, 12, 16, 20, 24, 28, 32, 40):
                if K * t > 4096 * 24:
                    continue
                lat = min(e.gemm_time(t, t, K, mode, lower, 1, 10) for _ in range(3))
                big = min(e.gemm_time(t, t, K, mode, lower, 2, 10) for _ in range(3))
                fl = 2.0 * (t * 128) ** 2 * K * (0.5 if lower else 1.0)
                row = {"lower": lower, "mode": mode, "K": K, "t": t, "lat_us": round(1e3 * lat, 1), "big_us": round(1e3 * big, 1),
                       "lat_tf": round(fl / lat / 1e9, 1), "big_tf": round(fl / big / 1e9, 1)}
                out.append(row)
                print(row, file=sys.stderr, flush=True)
e.close()
print(json.dumps(out))
