"""K^-1 = L^-T L^-1 of the gradient half on 64 x 64 tiles (GPX_KINV_LAT_MAX = tile rows up to which) against the persistent
128 x 128 launch: fit-step time and gradient bits over N."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))


def child(N):
    import numpy as np
    import bench_inputs
    from gpax_amd import _lib
    X, y, Xn, p = bench_inputs.synthetic_problem(N, 2, 64, seed=N)
    e = _lib.Engine(0)
    e.set_train(X)
    e.factor(0, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    g = e.lml_grad()
    t = {}
    for name, st in [("potrf", _lib.STAGE_POTRF), ("fit_step", _lib.STAGE_FITSTEP)]:
        e.time_stage(st, 1)
        t[name] = float(np.median([e.time_stage(st, 1) for _ in range(9)]))
    e.close()
    print(json.dumps({"t": t, "grad": [float(v).hex() for v in np.concatenate([g[0], [g[1], g[2]]])]}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
        sys.exit(0)
    out = {}
    for N in [1024, 1536, 2048, 3072, 4096, 5120, 6144, 8192]:
        row = {}
        for mx in (0, 999):
            env = dict(os.environ, GPX_KINV_LAT_MAX=str(mx))
            r = subprocess.run([sys.executable, __file__, "child", str(N)], env=env, capture_output=True, text=True, timeout=300)
            row["lat" if mx else "persist"] = json.loads(r.stdout.strip().splitlines()[-1])
        a, b = row["persist"], row["lat"]
        out[str(N)] = {"fit_step_ms_persist": a["t"]["fit_step"], "fit_step_ms_lat": b["t"]["fit_step"], "potrf_ms": a["t"]["potrf"],
                       "same_gradient_bits": a["grad"] == b["grad"]}
        print(N, out[str(N)], file=sys.stderr, flush=True)
    print(json.dumps(out))
