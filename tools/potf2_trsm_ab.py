"""A/B on one box, one context: the panel TRSM as a launch of its own ("nofuse") against riding in the potf2 launch
("fuse", potf2.hip potf2_trsm_kernel) — potrf, fit step and predict at the mid sizes whose chain it shortens.
Alternating rounds; device time between HIP events (gpx_time_stage), median over rounds."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
import bench_inputs

eng = _lib.Engine(0)
out = {}
for N, d in [(512, 1), (1024, 2), (2048, 2), (4096, 2), (5120, 2)]:
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, 1024 if N >= 2048 else 100, seed=0)
    eng.set_train(X)
    eng.factor(0, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    eng.lml_grad()
    eng.factor(0, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    eng.posterior(Xn, p["noise"], 1e-6, want_cov=True, want_var=False)
    eng.mvn_draw(np.zeros((1, Xn.shape[0])))
    rec = {m: {"potrf": [], "fit_step": [], "predict": []} for m in ("nofuse", "fuse")}
    for rnd in range(5):
        for mode in ("nofuse", "fuse"):
            eng.set_potf2(mode)
            for name, st in (("potrf", _lib.STAGE_POTRF), ("fit_step", _lib.STAGE_FITSTEP), ("predict", _lib.STAGE_PREDICT)):
                eng.time_stage(st, 2)
                rec[mode][name].append(eng.time_stage(st, 10) / 10)
    out[N] = {m: {k: float(np.median(v)) for k, v in r.items()} for m, r in rec.items()}
    a, b = out[N]["nofuse"], out[N]["fuse"]
    print(f"N={N}: potrf {a['potrf']:.3f} -> {b['potrf']:.3f} ms, fit step {a['fit_step']:.3f} -> {b['fit_step']:.3f}, "
          f"predict {a['predict']:.3f} -> {b['predict']:.3f}", flush=True)
print(json.dumps(out))
