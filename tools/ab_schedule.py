"""Within-process A/B of the Cholesky schedule knobs (GPX_LAZY_GROUP / GPX_EARLY_DIAG / GPX_OUTER_TILES are read at
gpx_init): interleaved rounds of the single-context stages and of the 3-contexts-in-flight sweep at C3."""
import json
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.getcwd())
from bench_inputs import synthetic_problem, synthetic_theta_samples  # noqa: E402
from gpax_amd import _lib  # noqa: E402

CONFIGS = json.loads(os.environ.get("AB_CONFIGS", '[{"GPX_LAZY_GROUP": "1"}, {"GPX_LAZY_GROUP": "2"}]'))
N, d, M = 16384, 2, 1024
X, y, Xn, p = synthetic_problem(N, d, M, seed=0)
th = synthetic_theta_samples(64, d, seed=1)


def make(cfg, n):
    for k, v in cfg.items():
        os.environ[k] = v
    engs = [_lib.Engine(0) for _ in range(n)]
    for k in cfg:
        os.environ.pop(k)
    for e in engs:
        e.set_train(X)
        e.factor(1, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        e.posterior(Xn, p["noise"], 1e-6, want_cov=True)
        e.mvn_draw(np.zeros((1, M)))
    return engs


pools = [make(c, 3) for c in CONFIGS]


def sweep3(engs, steps=18):
    idx = np.arange(steps)
    parts = [idx[i::3] for i in range(3)]
    ev = [0.0] * 3

    def work(i):
        ev[i] = engs[i].sweep_resident(1, th["k_length"][parts[i]], th["k_scale"][parts[i]], th["noise"][parts[i]], False,
                                       1e-6, 1)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    import time
    t0 = time.perf_counter()
    [t.start() for t in ts]
    [t.join() for t in ts]
    return steps / (time.perf_counter() - t0)


res = [dict(cfg=c, potrf=[], predict=[], fit=[], value=[]) for c in CONFIGS]
for rnd in range(4):
    for i, engs in enumerate(pools):
        e = engs[0]
        res[i]["potrf"].append(e.time_stage(_lib.STAGE_POTRF, 2) / 2)
        res[i]["predict"].append(e.time_stage(_lib.STAGE_PREDICT, 2) / 2)
        res[i]["fit"].append(e.time_stage(_lib.STAGE_FITSTEP, 1))
        res[i]["value"].append(sweep3(engs))
for r in res:
    print(json.dumps({"cfg": r["cfg"], **{k: [round(float(np.median(v)), 3), round(float(np.min(v)), 3), round(float(np.max(v)), 3)]
                                          for k, v in r.items() if k != "cfg"}}))
