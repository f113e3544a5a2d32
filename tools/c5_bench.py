"""C5 (BASELINE.json configs[4]): viGP / viSparseGP, Matern, on the 512 x 512 image of BASELINE.md §3 (6.25 % of the
pixels measured: N = 16 316; 'random' inducing points at ratio 0.125: M_ind = 2039) on ONE GPU.  Prints one JSON record
(redirect it to profiles/<round>/c5_sparse.json):

  sparse step   one SVI step of viSparseGP = gpx_sgp_bound with its gradient (sparse_gp.py:62-114 + autodiff there)
  sparse fit    50 SVI steps through viSparseGP.fit (host Adam included)
  sparse pred   viSparseGP.predict_in_batches over all 262 144 pixels (sparse_gp.py:173-223)
  exact legs    viGP: 50 SVI steps (each one device fit step) and predict_in_batches over all pixels, one factorisation

Flop model (1 FMA = 2 flop; M = M_ind padded to 128, N padded to 128) — what the launches compute, triangular
structure counted where the kernels use it (the library's own per-launch `work`, read back through gpx_profile_*):
  forward   potrf(Kuu) M^3/3 + Luu^-T (tree) M^3/3 + W = Kfu Luu^-T (one GEMM against Luu^-1, k range cut at the column
            tile)  N M^2 + A = I + W^T W / s2  N M^2 + potrf(A) M^3/3                                = 2 N M^2 + M^3
  gradient  LA^-T M^3/3 + A^-1 = LA^-T LA^-1  M^3/3 + Tu H  M^3 (k ranges start at the triangle) + the lower triangle of
            the symmetric Tu H Tu^T  M^3/2 + Tu R  M^3 + W (Tu R)^T  2 N M^2                         = 2 N M^2 + 19/6 M^3
  posterior of Ms points: forward + V1 = Ksu Luu^-T  Ms M^2 + V2 = V1 LA^-T  Ms M^2 (+ O(Ms M) mean / variance)
The Gram builds (Kuu, Kfu, Ksu: 8 B written per entry) and the on-the-fly dK contractions are HBM / VALU work and
are not in the MFMA count."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from bench_inputs import synthetic_sparse_image  # noqa: E402
from gpax_amd import _lib, viGP, viSparseGP  # noqa: E402
from gpax_amd.utils import get_keys, initialize_inducing_points, preprocess_sparse_image  # noqa: E402

PEAK = 78.6e12
img, sparse = synthetic_sparse_image(512, 512, 0.0625, seed=3)
X, y, X_full = preprocess_sparse_image(sparse)
ybar = y.mean()
y = y - ybar
Xu = initialize_inducing_points(X, 0.125, "random", get_keys(0)[0])
N, Mi = X.shape[0], Xu.shape[0]
Mp, Ntp = (Mi + 127) // 128 * 128, (N + 127) // 128 * 128
eng = _lib.get_engine(0)
eng.set_train(X)
ell, scale, noise = [25.0, 25.0], 1.0, 1e-2
rec = {"config": "C5: viGP / viSparseGP Matern on the 512x512 image (BASELINE.json configs[4])", "N": N, "M_ind": Mi,
       "pixels": int(X_full.shape[0]), "theta": {"k_length": ell, "k_scale": scale, "noise": noise}}


def median_ms(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def counted_flops(fn):
    eng.profile_enable(True)
    eng.profile_reset()
    fn()
    tot, ms = 0.0, 0.0
    for cls in (_lib.PROF_GEMM_TRAILING, _lib.PROF_GEMM_OTHER, _lib.PROF_POTF2):
        _, m, w = eng.profile_read(cls)
        tot += w
        ms += m
    eng.profile_enable(False)
    return tot, ms


calls = [0]


def fresh_noise():
    """A noise value that differs in its last bits from call to call: the library keeps the forward pass of the previous
    call while the next one brings bit-identical inputs (predict_in_batches slices) — a timing loop must not ride on that."""
    calls[0] += 1
    return noise * (1.0 + 1e-13 * calls[0])


for want_grad, key in ((False, "bound"), (True, "bound_and_gradient")):
    f = lambda: eng.sgp_bound(1, ell, scale, fresh_noise(), 1e-6, Xu, y, want_grad)  # noqa: E731
    ms = median_ms(f)
    flops, kernel_ms = counted_flops(f)
    model = 2.0 * Ntp * Mp * Mp + 1.0 * Mp ** 3 + (2.0 * Ntp * Mp * Mp + 19.0 / 6.0 * Mp ** 3 if want_grad else 0.0)
    rec[key] = {"ms": ms, "mfma_flops_counted": flops, "mfma_flops_model": model, "tflops": flops / (ms * 1e-3) / 1e12,
                "frac_of_fp64_peak": flops / (ms * 1e-3) / PEAK, "mfma_kernel_ms_sum": kernel_ms}
b, info, g = eng.sgp_bound(1, ell, scale, noise, 1e-6, Xu, y, True)
rec["bound_value"], rec["info"] = b, info

# posterior of all pixels in device-sized slices (what viSparseGP.predict_in_batches does by default)
chunk = 65536
def post_all():  # noqa: E302
    out = []
    nz = fresh_noise()  # one forward pass per sweep over the image (first slice), reused by the other slices
    for s0 in range(0, X_full.shape[0], chunk):
        m, _, v, _ = eng.sgp_posterior(1, ell, scale, nz, 1e-6, Xu, y, X_full[s0:s0 + chunk], 0.0, False, True)
        out.append(m)
    return np.concatenate(out)
ms = median_ms(post_all, reps=3)  # noqa: E305
flops, kernel_ms = counted_flops(post_all)
mean = post_all()
rec["posterior_all_pixels"] = {"ms": ms, "slices": int(np.ceil(X_full.shape[0] / chunk)), "mfma_flops_counted": flops,
                               "tflops": flops / (ms * 1e-3) / 1e12, "frac_of_fp64_peak": flops / (ms * 1e-3) / PEAK,
                               "rmse_vs_true_image": float(np.sqrt(np.mean((mean + ybar - img.reshape(-1)) ** 2))),
                               "note": "host -> device copy of the pixel coordinates and device -> host copy of mean / variance included"}

# through the model API: 50 SVI steps, then predict_in_batches
_lib.set_engine(None)
ms_ = viSparseGP(2, "Matern")
t0 = time.perf_counter()
ms_.fit(get_keys(0)[0], X, y, inducing_points_ratio=0.125, num_steps=50, step_size=5e-3, progress_bar=False,
        print_summary=False)
t_fit = time.perf_counter() - t0
t0 = time.perf_counter()
m_s, v_s = ms_.predict_in_batches(get_keys(0)[1], X_full, batch_size=1000, noiseless=True)
t_pred = time.perf_counter() - t0
rec["viSparseGP_api"] = {"fit_50_steps_s": t_fit, "ms_per_svi_step": t_fit / 50 * 1e3, "predict_in_batches_all_pixels_s": t_pred,
                         "loss_first_last": [float(ms_.loss[0]), float(ms_.loss[-1])]}
mv = viGP(2, "Matern")
t0 = time.perf_counter()
mv.fit(get_keys(0)[0], X, y, num_steps=50, step_size=5e-2, progress_bar=False, print_summary=False)
t_fit = time.perf_counter() - t0
theta = {"k_length": np.array(ell), "k_scale": np.float64(scale), "noise": np.float64(noise)}
t0 = time.perf_counter()
m_e, v_e = mv.predict_in_batches(get_keys(0)[1], X_full, batch_size=1000, samples=theta, noiseless=True)
t_pred = time.perf_counter() - t0
fit_flops = 50.0 * (Ntp + 128) ** 3
rec["viGP_exact_api"] = {"fit_50_steps_s": t_fit, "ms_per_svi_step": t_fit / 50 * 1e3,
                         "fit_frac_of_fp64_peak": fit_flops / t_fit / PEAK,
                         "predict_in_batches_all_pixels_s": t_pred,
                         "rmse_vs_true_image": float(np.sqrt(np.mean((m_e + ybar - img.reshape(-1)) ** 2)))}
print(json.dumps(rec))
