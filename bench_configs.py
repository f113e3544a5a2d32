#!/usr/bin/env python
"""
bench_configs.py — the OTHER BASELINE.json configs on the driver's clock (VERDICT r4 "next" #1).

bench.py's headline is C3 (configs[2]).  This module times, in the same bench.py run and live on the same GPU, one record
per remaining config — each with a checksum the way `lml_check` works for the headline — and prints ONE JSON object:

  C1  configs[0]  gpax.ExactGP(1, 'RBF') on N = 512 synthetic 1-D: NUTS 200 + 200 and predict through the model API
                  (gpax/models/gp.py:166-220,351-399)
  C2  configs[1]  ExactGP RBF N = 4096, d = 2, M = 1024: potrf / fit step / posterior / predict stage times (device
                  events), their fractions of the fp64 MFMA peak on SURVEY 8(d)'s flop counts, and a batched sweep
  C4  configs[3]  the S = 1000 posterior predictive sweep at N = 8192, d = 3, M = 1024, n = 1 through ExactGP.predict
                  (host arrays in and out: gp.py:351-399), posteriors/s and the fraction on F_post = 2.61e11 per sample
  C5  configs[4]  viGP / viSparseGP Matern on the 512 x 512 image (vigp.py:77-185, sparse_gp.py:62-223): the sparse bound
                  and bound + gradient (ms, fraction of the peak on the flop the launches count), 50 SVI steps through
                  the API, all 262 144 pixels; the exact viGP SVI step and its all-pixels predict

bench.py runs it as a child process (`python bench_configs.py`) AFTER the CPU-baseline leg, under a time limit: a config
that fails costs the headline line a note, never its result.  Nothing here touches oracle/ — these are timings of the
product path; parity of every one of these configs is tests/test_gpu_parity_fullsize.py's business.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = 78.6e12  # fp64 MFMA, SURVEY.md 8(d)


def _median_ms(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def c1():
    import bench_inputs
    from gpax_amd import ExactGP, _lib
    from gpax_amd.utils import get_keys

    N, d, M = 512, 1, 100
    X, y, Xn, _ = bench_inputs.synthetic_problem(N, d, M, seed=0)
    k1, k2 = get_keys()
    m = ExactGP(d, "RBF")
    t0 = time.perf_counter()
    m.fit(k1, X, y, num_warmup=200, num_samples=200, progress_bar=False, print_summary=False)
    t1 = time.perf_counter()
    ym, ys = m.predict(k2, Xn, n=1)
    t2 = time.perf_counter()
    s = m.get_samples()
    nl = int(sum(int(np.sum(st["n_leapfrog"])) for st in m.mcmc.get_extra_fields()))
    truth = np.prod(np.sin(Xn + 0.3 * np.arange(d)), axis=1)
    # where gpax actually runs (every reference notebook fits N = 6 ... 40 points): one lml + gradient evaluation as the
    # samplers issue it (gpx_fit_batch, B = 1 — ONE kernel launch up to N = 128, csrc/fit_small.hip), host to host
    eng = _lib.get_engine()
    small = {}
    for n_small in (25, 100, 128, 512):
        Xs, ys_, _, p = bench_inputs.synthetic_problem(n_small, d, 4, seed=1)
        eng.set_train(Xs)
        kind = _lib.kernel_kind("RBF")
        args = (kind, np.asarray(p["k_length"], dtype=float)[None, :], [p["k_scale"]], [p["noise"]], 1e-6, ys_)
        small[f"fit_step_ms_N{n_small}"] = _median_ms(lambda: eng.fit_batch(*args), reps=50)  # noqa: B023
    # ... and the simpleGP notebook's own size through the model API: NUTS 200 + 200 at N = 25
    Xs, ys_, _, _ = bench_inputs.synthetic_problem(25, 1, 4, seed=1)
    m25 = ExactGP(1, "RBF")
    t25 = time.perf_counter()
    m25.fit(k1, Xs, ys_, num_warmup=200, num_samples=200, progress_bar=False, print_summary=False)
    small["nuts_200_200_N25_s"] = time.perf_counter() - t25
    small["nuts_N25_leapfrogs"] = int(sum(int(np.sum(st["n_leapfrog"])) for st in m25.mcmc.get_extra_fields()))
    return {"config": "C1: ExactGP(1, 'RBF') N=512 d=1, NUTS 200 + 200, predict M=100 n=1 (BASELINE.json configs[0])",
            "fit_s": t1 - t0, "predict_s": t2 - t1, "leapfrogs_in_sampling": nl,
            "posterior_means": {k: np.asarray(v).mean(axis=0).ravel().tolist() for k, v in s.items()},
            "rmse_vs_truth": float(np.sqrt(np.mean((ym - truth) ** 2))),
            "checksum": float(np.sum(ym)), "host_api_fit_step": small}


def c2():
    import bench_inputs
    from gpax_amd import _lib

    N, d, M = 4096, 2, 1024
    X, y, Xn, p = bench_inputs.synthetic_problem(N, d, M, seed=0)
    kind = _lib.kernel_kind("RBF")
    eng = _lib.Engine(0)
    eng.set_train(X)
    lml, info = eng.factor(kind, p["k_length"], p["k_scale"], p["noise"], 1e-6, y)
    eng.posterior(Xn, p["noise"], 1e-6, want_cov=True)
    eng.mvn_draw(np.random.default_rng(2).standard_normal((1, M)))
    stages = {}
    for name, st in [("gram", _lib.STAGE_GRAM), ("potrf", _lib.STAGE_POTRF), ("fit_step", _lib.STAGE_FITSTEP),
                     ("posterior", _lib.STAGE_POSTERIOR), ("predict", _lib.STAGE_PREDICT)]:
        eng.time_stage(st, 1)
        stages[name + "_ms"] = float(np.median([eng.time_stage(st, 1) for _ in range(9)]))
    post = N ** 3 / 3 + N * N * M + N * M * M + 2 * N * N + 2 * N * M
    frac = {"potrf": (N ** 3 / 3) / (stages["potrf_ms"] * 1e-3) / PEAK,
            "fit_step": N ** 3 / (stages["fit_step_ms"] * 1e-3) / PEAK,
            "posterior": post / (stages["posterior_ms"] * 1e-3) / PEAK,
            "predict": (post + M ** 3 / 3 + M * M) / (stages["predict_ms"] * 1e-3) / PEAK}
    # the batched sweep (the vmap as a grid dimension) at this size
    S = 240
    th = bench_inputs.synthetic_theta_samples(S, d, seed=1)
    eps = np.random.default_rng(2).standard_normal((S, 1, M))
    eng.predict_sweep(kind, th["k_length"][:60], th["k_scale"][:60], th["noise"][:60], y, Xn, False, 1e-6, eps[:60])
    t0 = time.perf_counter()
    res = eng.predict_sweep(kind, th["k_length"], th["k_scale"], th["noise"], y, Xn, False, 1e-6, eps)
    dt = time.perf_counter() - t0
    eng.close()
    return {"config": "C2: ExactGP RBF N=4096 d=2 M=1024 (BASELINE.json configs[1])", "stages": stages,
            "stages_frac_of_fp64_peak": frac, "sweep_S": S, "sweep_posteriors_per_s": S / dt,
            "sweep_frac_of_fp64_peak": S * (post + M ** 3 / 3 + M * M) / dt / PEAK,
            "sweep_nan_rows": int(np.isnan(res[1]).any(axis=(1, 2)).sum()), "checksum": float(lml), "info": int(info)}


def c4(S=1000):
    import bench_inputs
    from gpax_amd import ExactGP, _lib
    from gpax_amd.utils import get_keys

    N, d, M = 8192, 3, 1024
    X, y, Xn, _ = bench_inputs.synthetic_problem(N, d, M, seed=0)
    th = bench_inputs.synthetic_theta_samples(S, d, seed=1)
    m = ExactGP(d, "Matern")
    m.X_train, m.y_train = m._set_data(X, y)
    samples = {"k_length": th["k_length"], "k_scale": th["k_scale"], "noise": th["noise"]}
    _, k2 = get_keys()
    m.predict(k2, Xn, {k: v[:min(S, 48)] for k, v in samples.items()}, n=1)  # allocations at this shape, clocks
    t0 = time.perf_counter()
    ym, ys = m.predict(k2, Xn, samples, n=1)
    dt = time.perf_counter() - t0
    flop = N ** 3 / 3 + N * N * M + N * M * M + 2 * N * N + 2 * N * M + M ** 3 / 3 + M * M
    return {"config": f"C4: {S}-sample posterior predictive sweep N=8192 d=3 M=1024 n=1 through ExactGP.predict, host arrays in "
                      "and out (BASELINE.json configs[3])",
            "S": S, "seconds": dt, "posteriors_per_s": S / dt, "tflops": S * flop / dt / 1e12,
            "frac_of_fp64_peak": S * flop / dt / PEAK, "flop_per_posterior": flop,
            "batch": _lib.get_engine().sweep_stats()[2], "contexts": len(_lib.get_sweep_engines()),
            "nan_rows": int(np.isnan(ys).any(axis=(1, 2)).sum()), "checksum": float(np.sum(ym))}


def c5():
    from bench_inputs import synthetic_sparse_image
    from gpax_amd import _lib, viGP, viSparseGP
    from gpax_amd.utils import get_keys, initialize_inducing_points, preprocess_sparse_image

    img, sparse = synthetic_sparse_image(512, 512, 0.0625, seed=3)
    X, y, X_full = preprocess_sparse_image(sparse)
    ybar = y.mean()
    y = y - ybar
    Xu = initialize_inducing_points(X, 0.125, "random", get_keys(0)[0])
    N, Mi = X.shape[0], Xu.shape[0]
    Mp, Ntp = (Mi + 127) // 128 * 128, (N + 127) // 128 * 128
    _lib.set_engine(None)
    eng = _lib.get_engine(0)
    eng.set_train(X)
    ell, scale, noise = [25.0, 25.0], 1.0, 1e-2
    rec = {"config": "C5: viGP / viSparseGP Matern on the 512x512 image, 6.25 % of the pixels, M_ind = ratio 0.125 "
                     "(BASELINE.json configs[4])", "N": int(N), "M_ind": int(Mi), "pixels": int(X_full.shape[0])}
    calls = [0]

    def fresh_noise():  # the library keeps a forward pass whose inputs come back bit-identical: never time that
        calls[0] += 1
        return noise * (1.0 + 1e-13 * calls[0])

    def counted(fn):
        eng.profile_enable(True)
        eng.profile_reset()
        fn()
        tot = 0.0
        for cls in (_lib.PROF_GEMM_TRAILING, _lib.PROF_GEMM_OTHER, _lib.PROF_POTF2):
            tot += eng.profile_read(cls)[2]
        eng.profile_enable(False)
        return tot

    for want_grad, key in ((False, "sparse_bound"), (True, "sparse_bound_and_gradient")):
        f = lambda: eng.sgp_bound(1, ell, scale, fresh_noise(), 1e-6, Xu, y, want_grad)  # noqa: E731
        ms = _median_ms(f, reps=9)
        flops = counted(f)
        model = 2.0 * Ntp * Mp * Mp + 1.0 * Mp ** 3 + (2.0 * Ntp * Mp * Mp + 19.0 / 6.0 * Mp ** 3 if want_grad else 0.0)
        rec[key] = {"ms": ms, "mfma_flops_counted": flops, "mfma_flops_model": model,
                    "frac_of_fp64_peak": flops / (ms * 1e-3) / PEAK}
    b, info, _ = eng.sgp_bound(1, ell, scale, noise, 1e-6, Xu, y, True)
    rec["checksum"], rec["info"] = float(b), int(info)
    chunk = 65536

    def post_all():
        nz, out = fresh_noise(), []
        for s0 in range(0, X_full.shape[0], chunk):
            out.append(eng.sgp_posterior(1, ell, scale, nz, 1e-6, Xu, y, X_full[s0:s0 + chunk], 0.0, False, True)[0])
        return np.concatenate(out)

    ms = _median_ms(post_all, reps=3)
    mean = post_all()
    rec["sparse_posterior_all_pixels"] = {"ms": ms, "rmse_vs_true_image":
                                          float(np.sqrt(np.mean((mean + ybar - img.reshape(-1)) ** 2)))}
    _lib.set_engine(None)
    sp = viSparseGP(2, "Matern")
    t0 = time.perf_counter()
    sp.fit(get_keys(0)[0], X, y, inducing_points_ratio=0.125, num_steps=50, step_size=5e-3, progress_bar=False,
           print_summary=False)
    t_fit = time.perf_counter() - t0
    t0 = time.perf_counter()
    sp.predict_in_batches(get_keys(0)[1], X_full, batch_size=1000, noiseless=True)
    rec["viSparseGP_api"] = {"ms_per_svi_step": t_fit / 50 * 1e3, "fit_50_steps_s": t_fit,
                             "predict_in_batches_all_pixels_s": time.perf_counter() - t0,
                             "loss_first_last": [float(sp.loss[0]), float(sp.loss[-1])]}
    ex = viGP(2, "Matern")
    steps = 20
    t0 = time.perf_counter()
    ex.fit(get_keys(0)[0], X, y, num_steps=steps, step_size=5e-2, progress_bar=False, print_summary=False)
    t_fit = time.perf_counter() - t0
    theta = {"k_length": np.array(ell), "k_scale": np.float64(scale), "noise": np.float64(noise)}
    t0 = time.perf_counter()
    m_e, _ = ex.predict_in_batches(get_keys(0)[1], X_full, batch_size=1000, samples=theta, noiseless=True)
    t_pred = time.perf_counter() - t0
    rec["viGP_exact_api"] = {"svi_steps": steps, "ms_per_svi_step": t_fit / steps * 1e3,
                             "fit_frac_of_fp64_peak": steps * float(Ntp + 128) ** 3 / t_fit / PEAK,
                             "predict_in_batches_all_pixels_s": t_pred,
                             "rmse_vs_true_image": float(np.sqrt(np.mean((m_e + ybar - img.reshape(-1)) ** 2)))}
    return rec


def main():
    only = [x for x in sys.argv[1:] if not x.startswith("-")]
    out = {}
    t_all = time.perf_counter()
    for name, fn in (("C1", c1), ("C2", c2), ("C4", c4), ("C5", c5)):
        if only and name not in only:
            continue
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as ex:  # one config's failure is that config's note
            out[name] = {"error": f"{type(ex).__name__}: {ex}"[:400]}
        out[name]["record_wall_s"] = time.perf_counter() - t0
    out["wall_s"] = time.perf_counter() - t_all
    out["note"] = ("measured live in this bench.py run (child process, same GPU, after the headline's timed region and the "
                   "CPU-baseline leg); fractions are of the 78.6 TFLOP/s fp64 MFMA peak on SURVEY.md 8(d)'s flop counts")
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
