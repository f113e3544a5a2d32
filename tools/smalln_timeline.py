"""Kernel timeline of ONE fit step (lml + gradient: one NUTS leapfrog / SVI step) at a small N, from a rocprofv3
--kernel-trace database: every launch with its start, duration and the gap to the launch before it.
    rocprofv3 --kernel-trace -d /tmp/sn -- python tools/smalln_timeline.py run 512
    python tools/smalln_timeline.py show $(find /tmp/sn -name '*.db') [out.md]
(the sizes gpax is mostly used at — gpax_simpleGP.ipynb: N = 25 ... 512.)"""
import os
import sqlite3
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())


def run(N):
    from bench_inputs import synthetic_problem
    from gpax_amd import _lib
    X, y, Xn, p = synthetic_problem(N, 1, 64, seed=0)
    e = _lib.Engine(0)
    e.set_train(X)
    for i in range(30):
        e.factor(0, p["k_length"], p["k_scale"], p["noise"] * (1 + 1e-3 * i), 1e-6, y)
        e.lml_grad()
    e.synchronize()
    t0 = time.perf_counter()
    reps = 200
    for i in range(reps):
        e.fit_batch(0, np.asarray(p["k_length"]).reshape(1, -1), [p["k_scale"]], [p["noise"] * (1 + 1e-4 * i)], 1e-6, y)
    dt = (time.perf_counter() - t0) / reps
    print(f"N={N}: {dt * 1e3:.3f} ms per fit step through gpx_fit_batch (host wall clock, B = 1)")
    e.close()


def show(dbp, out=None):
    db = sqlite3.connect(dbp)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    # the last fit step: from the last gram_kernel to the end
    # (N <= 127: the whole step is ONE launch, fit_small_kernel — csrc/fit_small.hip)
    gi = [i for i, r in enumerate(rows) if "gram_kernel" in r[0] or "fit_small_kernel" in r[0]]
    seg = rows[gi[-1]:]
    t0 = seg[0][1]
    lines = [f"{len(seg)} launches, span {(seg[-1][2] - t0) / 1e3:.1f} us, kernel time {sum(b - a for _, a, b in seg) / 1e3:.1f} us",
             "", "| # | kernel | start us | dur us | gap before us |", "|---|---|---|---|---|"]
    prev = None
    for i, (n, a, b) in enumerate(seg):
        short = n.split("(")[0].replace("void ", "").replace("gpx::", "")[:60]
        gap = (a - prev) / 1e3 if prev is not None else 0.0
        lines.append(f"| {i} | `{short}` | {(a - t0) / 1e3:.1f} | {(b - a) / 1e3:.1f} | {gap:.1f} |")
        prev = b
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        show(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
