"""Soak of the round-3 paths: the L^-T tree of the fit step and the sparse-GP solves by inverse — random sizes (ragged
tile counts), batch sizes and conditioning; every configuration is run twice on a fresh pattern of earlier calls and
must be bit-identical run to run (a race shows up as a differing bit), and a sample of them is checked against the
oracle.  Prints one JSON record."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
from oracle import cpu_ref as ref
import bench_inputs

rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
eng = _lib.Engine(0)
R = int(os.environ.get("ROUNDS", "60"))
worst_rel, n_oracle, mismatches = 0.0, 0, 0
t0 = time.perf_counter()
for it in range(R):
    N = int(rng.integers(130, 3000))
    d = int(rng.integers(1, 4))
    B = int(rng.integers(1, 6))
    noise = float(10.0 ** rng.uniform(-4, 0))
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, 64, seed=it, noise=max(noise, 1e-3))
    ells = np.stack([p["k_length"] * f for f in rng.uniform(0.7, 1.4, B)])
    sc, nz = np.full(B, p["k_scale"]), np.full(B, noise)
    eng.set_train(X)
    a = eng.fit_batch(1, ells, sc, nz, 1e-6, y)
    eng.factor(1, ells[0], sc[0], nz[0], 1e-6, y)  # different launches in between
    g1 = eng.lml_grad()
    b = eng.fit_batch(1, ells, sc, nz, 1e-6, y)
    for u, v in zip(a, b):
        if not np.array_equal(np.asarray(u), np.asarray(v)):
            mismatches += 1
    if not np.array_equal(np.concatenate([g1[0], [g1[1], g1[2]]]), a[2][0]):
        mismatches += 1
    if N <= 900 and a[1][0] == 0:
        pp = dict(k_length=ells[0], k_scale=sc[0], noise=nz[0])
        ge, gs, gn, _ = ref.exactgp_log_likelihood_grad(X, y, pp, kernel="Matern", jitter=1e-6)
        want = np.concatenate([ge, [gs, gn]])
        worst_rel = max(worst_rel, float(np.max(np.abs(a[2][0] - want)) / np.max(np.abs(want))))
        n_oracle += 1
    # sparse: inducing points = a random subset
    Mi = int(rng.integers(20, max(21, min(N // 2, 900))))
    Xu = X[rng.choice(N, Mi, replace=False)] + 1e-3
    s1 = eng.sgp_bound(1, ells[0], sc[0], max(noise, 1e-3), 1e-6, Xu, y, True)
    q1 = eng.sgp_posterior(1, ells[0], sc[0], max(noise, 1e-3), 1e-6, Xu, y, Xn, 0.0, want_cov=False, want_var=True)
    s2 = eng.sgp_bound(1, ells[0], sc[0], max(noise, 1e-3) * (1 + 1e-15), 1e-6, Xu, y, True)  # forces a fresh forward pass
    s3 = eng.sgp_bound(1, ells[0], sc[0], max(noise, 1e-3), 1e-6, Xu, y, True)
    if s1[0] != s3[0] or any(not np.array_equal(np.asarray(s1[2][k]), np.asarray(s3[2][k])) for k in s1[2]):
        mismatches += 1
rec = dict(rounds=R, seconds=time.perf_counter() - t0, run_to_run_mismatches=mismatches, oracle_checks=n_oracle,
           worst_gradient_rel_err_vs_oracle=worst_rel)
print(json.dumps(rec))
sys.exit(1 if mismatches else 0)
