"""Stand-alone device time of the chain GEMMs of the blocked Cholesky (gpax/models/gp.py:160-164 underneath): the rank-128
inner update, the in-place panel TRSM and U1 at C3 / C2 / sparse-chain scale — round-5 latency shapes (lat_tile) against
the round-1 kernels (GPX_LAT_GEMM=r1), and the 128 x 128 throughput shape where it applies.  One JSON line
(redirect to profiles/<round>/lat_gemm.json)."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
from gpax_amd import _lib  # noqa: E402

eng = _lib.Engine(0)
PEAK = 78.6e12
cases = [  # name, tiles_m, tiles_n, K, mode, lower
    ("inner update C3 (125 x 3 tiles, K=128, lower)", 125, 3, 128, 1, True),
    ("inner update C3 tail (40 x 3, K=128, lower)", 40, 3, 128, 1, True),
    ("inner update C2 one-block (31 x 31, K=128, lower)", 31, 31, 128, 1, True),
    ("inner update C2 mid (16 x 16, K=128, lower)", 16, 16, 128, 1, True),
    ("inner update sparse chain (8 x 8, K=128, lower)", 8, 8, 128, 1, True),
    ("panel TRSM C3 (125 x 1, in place)", 125, 1, 128, 2, False),
    ("panel TRSM C2 (31 x 1, in place)", 31, 1, 128, 2, False),
    ("panel TRSM (8 x 1, in place)", 8, 1, 128, 2, False),
    ("U1 C3 (120 x 4, K=512, lower)", 120, 4, 512, 1, True),
    ("U1 C3 tail (40 x 4, K=512, lower)", 40, 4, 512, 1, True),
    ("tree level (8 x 8, K=1024, beta 0)", 8, 8, 1024, 0, False),
    ("tree level (4 x 4, K=512, beta 0)", 4, 4, 512, 0, False),
]
out = []
for name, tm, tn, K, mode, lower in cases:
    flop = 2.0 * tm * 128 * tn * 128 * K
    if lower:  # tiles on or below the diagonal of the LAST tn tile columns of a tm-row panel: all of them here (tm >= tn)
        flop = 2.0 * 128 * 128 * K * sum(min(tn, i + 1) for i in range(tm)) if tm <= tn else flop
    rec = {"case": name, "flop": flop}
    for kern in ("r1", "r5"):
        eng.set_lat_gemm(kern)
        ms = min(eng.gemm_time(tm, tn, K, mode, lower, 1, 30) for _ in range(3))
        rec[kern + "_us"] = ms * 1e3
        rec[kern + "_tflops"] = flop / (ms * 1e-3) / 1e12
    if mode != 2:
        ms = min(eng.gemm_time(tm, tn, K, mode, lower, 2, 30) for _ in range(3))
        rec["big_us"] = ms * 1e3
        rec["big_tflops"] = flop / (ms * 1e-3) / 1e12
    rec["r5_over_r1"] = rec["r1_us"] / rec["r5_us"]
    out.append(rec)
    print(f"{name:52s} r1 {rec['r1_us']:7.1f} us {rec['r1_tflops']:5.1f} TF | r5 {rec['r5_us']:7.1f} us {rec['r5_tflops']:5.1f} TF"
          + (f" | 128x128 {rec['big_us']:7.1f} us {rec['big_tflops']:5.1f} TF" if "big_us" in rec else ""), file=sys.stderr, flush=True)
print(json.dumps({"peak_tflops": PEAK / 1e12, "note": "back-to-back launches of one GEMM on resident constant operands, HIP events, "
                  "min of 3 x 30 launches; lower: tiles above the diagonal return at once", "cases": out}))
