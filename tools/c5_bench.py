"""C5 (BASELINE.json configs[4]): viSparseGP Matern, 512x512 image, 6.25 % of pixels observed (N ~ 16384),
M_ind = 2048: time of one SVI step (VFE bound + gradient) and of the posterior over all 262 144 pixels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from gpax_amd import _lib
eng = _lib.Engine(0)
rng = np.random.default_rng(3)
H = W = 512
ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
img = 1.5 + np.sin(ii / 40.0) * np.cos(jj / 55.0)
keep = rng.uniform(size=img.shape) < 0.0625
X = np.column_stack(np.nonzero(keep)).astype(np.float64)
y = img[keep] - img[keep].mean()
N, Mi = X.shape[0], 2048
Xu = X[rng.choice(N, Mi, replace=False)]
Xs = np.column_stack([ii.reshape(-1), jj.reshape(-1)]).astype(np.float64)
eng.set_train(X)
ell, scale, noise = [30.0, 30.0], 1.0, 1e-2
for want_grad in (False, True):
    eng.sgp_bound(1, ell, scale, noise, 1e-6, Xu, y, want_grad)
    t0 = time.perf_counter()
    for _ in range(5):
        b, info, g = eng.sgp_bound(1, ell, scale, noise, 1e-6, Xu, y, want_grad)
    dt = (time.perf_counter() - t0) / 5
    print(f"N={N} M_ind={Mi}: sgp_bound(grad={want_grad}) {dt*1e3:.1f} ms  (bound {b:.3f}, info {info})", flush=True)
for chunk in (1000, 16384, 65536):
    eng.sgp_posterior(1, ell, scale, noise, 1e-6, Xu, y, Xs[:chunk], noise, False, True)
    t0 = time.perf_counter()
    out = []
    for s0 in range(0, Xs.shape[0], chunk):
        m, _, v, info = eng.sgp_posterior(1, ell, scale, noise, 1e-6, Xu, y, Xs[s0:s0 + chunk], noise, False, True)
        out.append(m)
    dt = time.perf_counter() - t0
    mean = np.concatenate(out)
    rmse = np.sqrt(np.mean((mean + img[keep].mean() - img.reshape(-1)) ** 2))
    print(f"posterior over {Xs.shape[0]} pixels in slices of {chunk}: {dt*1e3:.0f} ms, rmse {rmse:.4f}", flush=True)
