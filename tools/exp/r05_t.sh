cd $GRAFT_REPO_ROOT
for ot in 1 2 3 4; do
GPX_ONE_STREAM_OT=$ot timeout 300 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
import bench_inputs
from gpax_amd import _lib
out = []
for N in (1024, 2048, 3072, 4096, 5120):
    X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, 1024, seed=0)
    e = _lib.Engine(0); e.set_train(X)
    lml, info = e.factor(0, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    e.posterior(Xn, p["noise"], 1e-6, want_cov=True)
    t = {}
    for name, st in (("potrf", _lib.STAGE_POTRF), ("fit", _lib.STAGE_FITSTEP), ("predict", _lib.STAGE_PREDICT)):
        e.time_stage(st, 1)
        t[name] = float(np.median([e.time_stage(st, 1) for _ in range(9)]))
    out.append("N=%d potrf %.3f fit %.3f predict %.3f lml %.9f" % (N, t["potrf"], t["fit"], t["predict"], lml))
    e.close()
print("OT", os.environ["GPX_ONE_STREAM_OT"], " | ".join(out))
PY
done
