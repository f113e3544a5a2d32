cd $GRAFT_REPO_ROOT
for ed in 0 8 16 24; do
GPX_EARLY_DIAG_MIN=$ed timeout 300 python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
import bench_inputs
from gpax_amd import _lib
out = []
for N in (1024, 2048, 3072, 4096, 5120):
    X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, 1024, seed=0)
    e = _lib.Engine(0); e.set_train(X)
    lml, info = e.factor(0, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    mean, cov, _ = e.posterior(Xn, p["noise"], 1e-6, want_cov=True)
    t = {}
    for name, st in (("potrf", _lib.STAGE_POTRF), ("fit", _lib.STAGE_FITSTEP), ("predict", _lib.STAGE_PREDICT)):
        e.time_stage(st, 1)
        t[name] = float(np.median([e.time_stage(st, 1) for _ in range(9)]))
    out.append("N=%d potrf %.3f fit %.3f predict %.3f lml %.10f cs %.10f" % (N, t["potrf"], t["fit"], t["predict"], lml, float(np.sum(cov))))
    e.close()
print("EARLY_DIAG_MIN", os.environ["GPX_EARLY_DIAG_MIN"], " | ".join(out))
PY
done
