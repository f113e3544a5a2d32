"""Mint the golden fixtures under tests/golden/ from oracle/cpu_ref.py (seeded, small).

The reference itself cannot be imported here (no jax / numpyro), so these vectors are produced by
the CPU restatement; tests/test_oracle.py cross-checks the same quantities independently
(mpmath, scipy.stats, finite differences, explicit-inverse route).  Re-running this script must
reproduce the committed files bit for bit (checked by tests/test_oracle.py::test_golden_reproducible).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_ref as ref  # noqa: E402
import bench_inputs

OUT = os.path.join(ROOT, "tests", "golden")


def gram_cases():
    out = {}
    rng = np.random.default_rng(100)
    cid = 0
    for name in ["RBF", "Matern"]:
        for (n, m, d) in [(5, 5, 1), (64, 64, 2), (64, 37, 2), (130, 67, 3)]:
            X = rng.uniform(0, 10, (n, d))
            Z = X.copy() if n == m else rng.uniform(0, 10, (m, d))
            if n == m:
                X[1] = X[0]  # coincident points exercise clip / sqrt eps
                Z = X.copy()
            for ard in [False, True]:
                ell = (0.7 + rng.uniform(0, 2, d)) if ard else np.array(1.3)
                for noise in [0.0, 0.25]:
                    p = {"k_length": ell, "k_scale": 1.7}
                    K = ref.get_kernel(name)(X, Z, p, noise=noise, jitter=1e-6)
                    out[f"c{cid}_X"], out[f"c{cid}_Z"], out[f"c{cid}_ell"] = X, Z, np.asarray(ell, dtype=np.float64)
                    out[f"c{cid}_meta"] = np.array([0 if name == "RBF" else 1, 1.7, noise, 1e-6])
                    out[f"c{cid}_K"] = K
                    cid += 1
    out["ncases"] = np.array(cid)
    return out


def lml_cases():
    out = {}
    cid = 0
    for name in ["RBF", "Matern"]:
        for (N, d) in [(64, 1), (200, 2), (512, 3)]:
            X, y, _, p = bench_inputs.synthetic_problem(N, d, 4, seed=N + d)
            for k in range(2):
                rng = np.random.default_rng(7 * N + k)
                p2 = {"k_length": p["k_length"] * np.exp(0.3 * rng.standard_normal(d)),
                      "k_scale": p["k_scale"] * float(np.exp(0.3 * rng.standard_normal())),
                      "noise": p["noise"] * float(np.exp(0.3 * rng.standard_normal()))}
                lml = ref.exactgp_log_likelihood(X, y, p2, kernel=name)
                g_ell, g_s, g_n, alpha = ref.exactgp_log_likelihood_grad(X, y, p2, kernel=name)
                out[f"c{cid}_X"], out[f"c{cid}_y"] = X, y
                out[f"c{cid}_theta"] = np.concatenate([p2["k_length"], [p2["k_scale"], p2["noise"]]])
                out[f"c{cid}_meta"] = np.array([0 if name == "RBF" else 1, 1e-6])
                out[f"c{cid}_lml"] = np.array(lml)
                out[f"c{cid}_grad"] = np.concatenate([g_ell, [g_s, g_n]])
                out[f"c{cid}_alpha"] = alpha
                cid += 1
    out["ncases"] = np.array(cid)
    return out


def posterior_cases():
    out = {}
    cid = 0
    for name in ["RBF", "Matern"]:
        for (N, d, M) in [(50, 1, 20), (128, 2, 64), (300, 3, 70)]:
            X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=N + M)
            for noiseless in [0, 1]:
                for jitter in [1e-6, 1e-5]:
                    mean, cov = ref.get_mvn_posterior(X, y, Xn, p, bool(noiseless), kernel=name, jitter=jitter,
                                                      route="inv")
                    eps = np.random.default_rng(cid).standard_normal((2, M))
                    draws = ref.mvn_sample(mean, cov, eps)
                    out[f"c{cid}_X"], out[f"c{cid}_y"], out[f"c{cid}_Xn"] = X, y, Xn
                    out[f"c{cid}_theta"] = np.concatenate([p["k_length"], [p["k_scale"], p["noise"]]])
                    out[f"c{cid}_meta"] = np.array([0 if name == "RBF" else 1, noiseless, jitter])
                    out[f"c{cid}_mean"], out[f"c{cid}_cov"] = mean, cov
                    out[f"c{cid}_eps"], out[f"c{cid}_draws"] = eps, draws
                    cid += 1
    out["ncases"] = np.array(cid)
    return out


def sweep_case():
    N, d, M, S, n = 200, 3, 48, 16, 2
    X, y, Xn, _ = bench_inputs.synthetic_problem(N, d, M, seed=21)
    samples = bench_inputs.synthetic_theta_samples(S, d, seed=1)
    eps = np.random.default_rng(2).standard_normal((S, n, M))
    out = {"X": X, "y": y, "Xn": Xn, "eps": eps, **{f"s_{k}": v for k, v in samples.items()}}
    for name in ["RBF", "Matern"]:
        mm, yy, means = ref.predict(X, y, Xn, samples, eps, False, kernel=name, route="inv")
        out[f"{name}_mean_of_means"], out[f"{name}_y_sampled"], out[f"{name}_means"] = mm, yy, means
    return out


def sparse_cases():
    out = {}
    N, d, M, Mi = 300, 2, 40, 30
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=9)
    Xu = X[np.random.default_rng(3).choice(N, Mi, replace=False)]
    out.update({"X": X, "y": y, "Xn": Xn, "Xu": Xu,
                "theta": np.concatenate([p["k_length"], [p["k_scale"], p["noise"]]])})
    for name in ["RBF", "Matern"]:
        out[f"{name}_bound"] = np.array(ref.sparse_bound(X, y, Xu, p, kernel=name))
        for noiseless in [0, 1]:
            mean, cov = ref.sparse_posterior(X, y, Xu, Xn, p, bool(noiseless), kernel=name)
            out[f"{name}_mean_{noiseless}"], out[f"{name}_cov_{noiseless}"] = mean, cov
    return out


def utils_cases():
    out = {}
    img = np.zeros((3, 4))
    img[0, 1], img[2, 3], img[1, 0] = 1.5, -2.0, 0.25
    a, b, c = ref.preprocess_sparse_image(img)
    out.update({"img34": img, "img34_X": a, "img34_y": b, "img34_full": c})
    rng = np.random.default_rng(4)
    img2 = rng.uniform(0.1, 1, (16, 16)) * (rng.uniform(size=(16, 16)) < 0.3)
    a, b, c = ref.preprocess_sparse_image(img2)
    out.update({"img16": img2, "img16_X": a, "img16_y": b, "img16_full": c})
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in [("gram", gram_cases), ("lml", lml_cases), ("posterior", posterior_cases),
                     ("sweep", sweep_case), ("sparse", sparse_cases), ("utils", utils_cases)]:
        np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fn())
        print("wrote", name)


if __name__ == "__main__":
    main()
