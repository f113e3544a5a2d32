"""A library switch off (VAR=0) against on (VAR=1) in one process: the same bits, and the stage times of a single-sample
factorisation / fit step / posterior over N.  Usage: env_ab.py VAR [N ...].  One JSON object on stdout."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench_inputs  # noqa: E402
from gpax_amd import _lib  # noqa: E402


def run(N, on, M=256, reps=9):
    os.environ[VAR] = "1" if on else "0"
    X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, M, seed=N)
    e = _lib.Engine(0)
    e.set_train(X)
    lml, info = e.factor(0, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    g = e.lml_grad()
    e.factor(0, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    mean, cov, _ = e.posterior(Xn, p["noise"], 1e-6, want_cov=True)
    t = {}
    for name, st in [("potrf", _lib.STAGE_POTRF), ("fit_step", _lib.STAGE_FITSTEP), ("posterior", _lib.STAGE_POSTERIOR)]:
        e.time_stage(st, 1)
        t[name] = float(np.median([e.time_stage(st, 1) for _ in range(reps)]))
    e.close()
    return dict(lml=lml, info=info, grad=np.concatenate([g[0], [g[1], g[2]]]), alpha=g[3], mean=mean, cov=cov, t=t)


VAR = sys.argv[1]


def main():
    out = {}
    for N in [int(a) for a in sys.argv[2:]] or [200, 512, 1000, 2048, 3000, 4096, 5120]:
        a, b = run(N, False), run(N, True)
        same = (a["lml"] == b["lml"] and a["info"] == b["info"] and np.array_equal(a["grad"], b["grad"]) and
                np.array_equal(a["alpha"], b["alpha"]) and np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["cov"], b["cov"]))
        out[str(N)] = {"same_bits": bool(same), "ms_off": a["t"], "ms_on": b["t"], "lml": a["lml"]}
        print(N, same, a["t"], b["t"], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
