"""k slabs for the triangular products of the gradient half (GPX_KSLAB="kchunk,min_nt,max_nt") against one launch:
fit-step time and gradient / alpha bits over N.  Usage: kslab_ab.py [kchunk ...]"""
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
    print(json.dumps({"t": t, "grad": [float(v).hex() for v in np.concatenate([g[0], [g[1], g[2]], g[3][:: max(1, N // 64)]])]}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
        sys.exit(0)
    kcs = [int(a) for a in sys.argv[1:]] or [512, 1024]
    out = {}
    for N in [1024, 1536, 2048, 3072, 4096, 5120, 6144]:
        row = {}
        for kc in [0] + kcs:
            env = dict(os.environ)
            if kc:
                env["GPX_KSLAB"] = f"{kc},1,999"
            else:
                env.pop("GPX_KSLAB", None)
            r = subprocess.run([sys.executable, __file__, "child", str(N)], env=env, capture_output=True, text=True, timeout=300)
            row[kc] = json.loads(r.stdout.strip().splitlines()[-1])
        out[str(N)] = {"potrf_ms": row[0]["t"]["potrf"],
                       "fit_step_ms": {str(k): round(v["t"]["fit_step"], 4) for k, v in row.items()},
                       "same_bits": {str(k): v["grad"] == row[0]["grad"] for k, v in row.items() if k}}
        print(N, out[str(N)], file=sys.stderr, flush=True)
    print(json.dumps(out))
