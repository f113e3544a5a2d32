cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 300 python - <<'PY'
import numpy as np, sys, time
sys.path.insert(0, '.')
import bench_inputs
from gpax_amd import _lib
from oracle import cpu_ref as ref
import os
for N, d, kind, name in [(25, 1, 0, "RBF"), (40, 2, 1, "Matern"), (127, 3, 0, "RBF"), (100, 5, 1, "Matern"), (7, 1, 1, "Matern"), (64, 2, 0, "RBF")]:
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, 9, seed=N)
    res = {}
    for mode in ("1", "0"):
        os.environ["GPX_FIT_SMALL"] = mode
        e = _lib.Engine(0)
        e.set_train(X)
        lml, info = e.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        g = e.lml_grad()
        lml2, info2 = e.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
        mean, cov, _ = e.posterior(Xn, p["noise"], 1e-6)
        fb = e.fit_batch(kind, np.stack([p["k_length"]] * 3) * np.array([[1.0], [1.1], [0.9]]), [p["k_scale"]] * 3, [p["noise"]] * 3, 1e-6, y)
        t0 = time.perf_counter()
        for _ in range(200):
            e.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y); e.lml_grad()
        dt = (time.perf_counter() - t0) / 200
        res[mode] = (lml, info, g, mean, cov, fb, dt)
        e.close()
    a, b = res["1"], res["0"]
    expect = ref.exactgp_log_likelihood(X, y, p, kernel=name)
    e_ell, e_scale, e_noise, _ = ref.exactgp_log_likelihood_grad(X, y, p, kernel=name)
    print(f"N={N} d={d} {name}: lml fused {a[0]:.12f} general {b[0]:.12f} oracle {expect:.12f} info {a[1]} {b[1]}")
    print("   grad ell  fused", a[2][0], "general", b[2][0], "oracle", e_ell)
    print("   scale/noise fused", a[2][1], a[2][2], "general", b[2][1], b[2][2], "oracle", e_scale, e_noise)
    print("   alpha maxdiff", np.abs(a[2][3] - b[2][3]).max(), " mean bitwise", np.array_equal(a[3], b[3]), "cov bitwise", np.array_equal(a[4], b[4]))
    print("   fit_batch lml", a[5][0], b[5][0], " single==batch[0]:", a[5][0][0] == a[0], np.array_equal(a[5][2][0][:d], a[2][0]))
    print(f"   host fit step: fused {a[6]*1e6:.1f} us  general {b[6]*1e6:.1f} us")
PY
