"""A checker-backed stand-in for gpax_amd._lib.Engine, used ONLY by the CPU tests of the host-side
logic (NUTS / SVI drivers, model plumbing, sharding).  It implements the Engine methods with
oracle/cpu_ref.py so those code paths can run where no GPU exists.  It lives under tests/ on
purpose: the product has no CPU fallback."""
import numpy as np
import scipy.linalg as sla

from gpax_amd._lib import broadcast_lengthscale
from oracle import cpu_ref as ref

_NAMES = {0: "RBF", 1: "Matern", 2: "Periodic"}


def _params(kind, ell, d, scale, noise=None):
    ell = np.asarray(ell, dtype=np.float64).reshape(-1)
    p = {"k_length": broadcast_lengthscale(ell[:d] if kind == 2 else ell, d), "k_scale": float(scale)}
    if kind == 2:
        p["period"] = float(ell[d])
    if noise is not None:
        p["noise"] = float(noise)
    return p


class OracleEngine:
    device = 0

    def __init__(self):
        self.N = self.d = self.M = 0
        self._post = None

    def set_train(self, X):
        X = np.asarray(X, dtype=np.float64)
        self._Xt = X if X.ndim == 3 else None  # (T, N, d): task-specific training sets (vExactGP)
        self.T = X.shape[0] if X.ndim == 3 else 1
        self.X = X[0] if X.ndim == 3 else X
        self.N, self.d = self.X.shape
        self._diag = None
        self._diag_key = None
        self._train_owner = None  # as _lib.Engine.set_train: the previous owner's residency claim ends here
        self._train_version = None

    def set_diag(self, v):
        self._diag = None if v is None else np.asarray(v, dtype=np.float64).reshape(-1)

    def _row(self, arr, b, count):
        """Row of a (N,) | (count, N) | (T, N) table for batch entry b."""
        arr = np.asarray(arr, dtype=np.float64)
        if arr.ndim == 1:
            return arr
        return arr[b] if arr.shape[0] == count else arr[b % self.T]

    def gram(self, kind, X, Z, ell, scale, diag_add, add_diag):
        X, Z = np.asarray(X, dtype=np.float64), np.asarray(Z, dtype=np.float64)
        if kind == 3:  # GPX_KERNEL_R2: the squared scaled distance itself
            return ref.square_scaled_distance(X, Z, np.asarray(ell, dtype=np.float64).reshape(-1))
        p = _params(kind, ell, X.shape[1], scale)
        K = ref.get_kernel(_NAMES[kind])(X, Z, p, noise=0.0, jitter=0.0)
        if add_diag:
            K = K + diag_add * np.eye(X.shape[0])
        return K

    def potrf(self, A):
        try:
            return np.linalg.cholesky(A), 0
        except np.linalg.LinAlgError:
            return np.full_like(A, np.nan), 1

    def factor(self, kind, ell, scale, noise, jitter, yres):
        self._theta = dict(kind=kind, p=_params(kind, ell, self.d, scale, noise), jitter=float(jitter), ell=ell,
                           scale=float(scale), noise=float(noise))
        self._yres = np.asarray(yres, dtype=np.float64)
        K = self.gram(kind, self.X, self.X, ell, scale, noise + jitter, True)
        if getattr(self, "_diag", None) is not None:
            K = K + np.diag(self._diag)
        try:
            if not np.all(np.isfinite(K)):  # extreme theta: the device kernel reports a failed pivot here too
                raise np.linalg.LinAlgError("non-finite Gram matrix")
            self._L = np.linalg.cholesky(K)
            if not np.all(np.isfinite(self._L)):
                raise np.linalg.LinAlgError("non-finite factor")
        except np.linalg.LinAlgError:
            self._L = None
            return float("nan"), 1
        if not np.all(np.isfinite(self._yres)):  # a mean function overflowed: the device returns NaN for the lml as well
            self._w = np.full(self.N, np.nan)
            return float("nan"), 0
        w = sla.solve_triangular(self._L, self._yres, lower=True)
        self._w = w
        lml = -0.5 * w @ w - np.log(np.diag(self._L)).sum() - 0.5 * self.N * ref.LOG_2PI
        return float(lml), 0

    def lml_grad(self):
        t = self._theta
        if t["kind"] != 2:
            return ref.exactgp_log_likelihood_grad(self.X, self._yres, t["p"], kernel=_NAMES[t["kind"]],
                                                   jitter=t["jitter"], yres=self._yres,
                                                   measured_noise=getattr(self, "_diag", None))
        # periodic: central differences of the oracle lml (tests only)
        ell = np.asarray(t["ell"], dtype=np.float64).reshape(-1)

        def f(e, s, n):
            return ref.exactgp_log_likelihood(self.X, self._yres, _params(2, e, self.d, s, n), kernel="Periodic",
                                              jitter=t["jitter"], measured_noise=getattr(self, "_diag", None))

        g = np.empty(ell.size)
        for m in range(ell.size):
            h = 1e-6 * ell[m]
            a, b = ell.copy(), ell.copy()
            a[m] += h
            b[m] -= h
            g[m] = (f(a, t["scale"], t["noise"]) - f(b, t["scale"], t["noise"])) / (2 * h)
        hs, hn = 1e-6 * t["scale"], 1e-6 * t["noise"]
        gs = (f(ell, t["scale"] + hs, t["noise"]) - f(ell, t["scale"] - hs, t["noise"])) / (2 * hs)
        gn = (f(ell, t["scale"], t["noise"] + hn) - f(ell, t["scale"], t["noise"] - hn)) / (2 * hn)
        K = ref.PeriodicKernel(self.X, self.X, t["p"], t["noise"], jitter=t["jitter"])
        if getattr(self, "_diag", None) is not None:
            K = K + np.diag(self._diag)
        return g, gs, gn, np.linalg.solve(K, self._yres)

    def lml_grad_diag(self):
        t = self._theta
        K = self.gram(t["kind"], self.X, self.X, t["ell"], t["scale"], t["noise"] + t["jitter"], True)
        if getattr(self, "_diag", None) is not None:
            K = K + np.diag(self._diag)
        Kinv = np.linalg.inv(K)
        a = Kinv @ self._yres
        return 0.5 * (a * a - np.diag(Kinv))

    def fit_batch(self, kind, ells, scales, noises, jitter, yres, want_grad=True):
        ells = np.asarray(ells, dtype=np.float64)
        B = ells.shape[0]
        yres = np.asarray(yres, dtype=np.float64)
        lml, info = np.empty(B), np.zeros(B, dtype=np.int32)
        grad, alpha = ([None] * B, [None] * B)
        for b in range(B):
            if self._Xt is not None:
                self.X = self._Xt[b % self.T]
            lml[b], info[b] = self.factor(kind, ells[b], scales[b], noises[b], jitter, self._row(yres, b, B))
            if want_grad and info[b] == 0:
                g_ell, gs, gn, alpha[b] = self.lml_grad()
                grad[b] = np.concatenate([np.asarray(g_ell).reshape(-1), [gs, gn]])
        if not want_grad:
            return lml, info, None, None
        ne = ells.shape[1] + 2
        grad = np.stack([g if g is not None else np.full(ne, np.nan) for g in grad])
        alpha = np.stack([a if a is not None else np.full(self.N, np.nan) for a in alpha])
        return lml, info, grad, alpha

    def posterior(self, Xnew, noise_p, jitter, want_cov=True, want_var=False):
        t = self._theta
        Xnew = np.asarray(Xnew, dtype=np.float64)
        self.M = Xnew.shape[0]
        if self._L is None:
            nanv = np.full(self.M, np.nan)
            self._post = (nanv, np.full((self.M, self.M), np.nan))
            return nanv, (np.full((self.M, self.M), np.nan) if want_cov else None), (nanv if want_var else None)
        kfn = ref.get_kernel(_NAMES[t["kind"]])
        k_pX = kfn(Xnew, self.X, t["p"], jitter=0.0)
        if Xnew.shape == self.X.shape:  # the reference's shape rule would add a diagonal here; k_pX never has one
            k_pX = k_pX - 0.0 * np.eye(self.N)
        V = sla.solve_triangular(self._L, k_pX.T, lower=True)
        mean = V.T @ self._w
        k_pp = kfn(Xnew, Xnew, t["p"], noise_p, jitter=jitter)
        cov = k_pp - V.T @ V
        self._post = (mean, cov)
        return mean, (cov if want_cov else None), (np.diag(cov).copy() if want_var else None)

    def mvn_draw(self, eps):
        mean, cov = self._post
        if np.isnan(cov).any():
            return np.full((np.asarray(eps).shape[0], mean.shape[0]), np.nan), 1
        out = ref.mvn_sample(mean, cov, np.asarray(eps))
        return out, int(np.isnan(out).any())

    def predict_sweep(self, kind, ells, scales, noises, yres, Xnew, noiseless, jitter, eps, want_var=False,
                      pred_diag=None, m_slice=0):
        Xnew = np.asarray(Xnew, dtype=np.float64)
        Mtot = Xnew.shape[-2]
        if 0 < m_slice < Mtot:  # the reference's predict_in_batches: an independent predict per slice of X_new
            outs = []
            for m0 in range(0, Mtot, m_slice):
                sl = slice(m0, min(m0 + m_slice, Mtot))
                outs.append(self.predict_sweep(kind, ells, scales, noises, yres, Xnew[..., sl, :], noiseless, jitter,
                                               None if eps is None else np.asarray(eps)[..., sl], True,
                                               None if pred_diag is None else np.asarray(pred_diag)[:, sl]))
            means = np.concatenate([o[0] for o in outs], axis=-1)
            samples = np.concatenate([o[1] for o in outs], axis=-1)
            infos = np.zeros(means.shape[0], dtype=np.int32)
            for o in outs:
                infos = np.where(infos != 0, infos, o[2])
            vars_ = np.concatenate([o[3] for o in outs], axis=-1)
            return (means, samples, infos, vars_) if want_var else (means, samples, infos)
        ells = np.asarray(ells, dtype=np.float64)
        S = ells.shape[0]
        Xnew = np.asarray(Xnew, dtype=np.float64)
        M = Xnew.shape[-2]
        n = 0 if eps is None else np.asarray(eps).shape[1]
        yres = np.asarray(yres, dtype=np.float64)
        means, samples, infos = np.empty((S, M)), np.empty((S, n, M)), np.zeros(S, dtype=np.int32)
        vars_ = np.empty((S, M))
        for s in range(S):
            if self._Xt is not None:
                self.X = self._Xt[s % self.T]
            lml, info = self.factor(kind, ells[s], scales[s], noises[s], jitter, self._row(yres, s, S))
            m, cov, _ = self.posterior(Xnew if Xnew.ndim == 2 else Xnew[s % self.T], 0.0 if noiseless else noises[s],
                                       jitter)
            if pred_diag is not None and self._post is not None and np.all(np.isfinite(cov)):
                cov = cov + np.diag(np.asarray(pred_diag, dtype=np.float64)[s])
                self._post = (m, cov)
            means[s] = m
            vars_[s] = np.diag(cov)
            infos[s] = info
            if n:
                samples[s], bad = self.mvn_draw(eps[s])
                if bad and not info:
                    infos[s] = -1
        if want_var:
            return means, samples, infos, vars_
        return means, samples, infos

    # ---- sparse GP (tests only; gradient by central differences of the oracle bound) -----------------
    def sgp_bound(self, kind, ell, scale, noise, jitter, Xu, yres, want_grad=True):
        name = _NAMES[kind]
        ell = broadcast_lengthscale(ell, self.d)
        Xu = np.asarray(Xu, dtype=np.float64)
        yres = np.asarray(yres, dtype=np.float64)

        def f(e, s, n, xu, yy=yres):
            try:
                return ref.sparse_bound(self.X, yy, xu, {"k_length": e, "k_scale": s, "noise": n}, kernel=name,
                                        jitter=jitter)
            except np.linalg.LinAlgError:
                return float("nan")

        b = f(ell, scale, noise, Xu)
        if not np.isfinite(b):
            return b, 1, None
        if not want_grad:
            return b, 0, None
        h = 1e-6
        g_ell = np.empty(self.d)
        for m in range(self.d):
            a, c = ell.copy(), ell.copy()
            a[m] += h
            c[m] -= h
            g_ell[m] = (f(a, scale, noise, Xu) - f(c, scale, noise, Xu)) / (2 * h)
        g_s = (f(ell, scale + h, noise, Xu) - f(ell, scale - h, noise, Xu)) / (2 * h)
        g_n = (f(ell, scale, noise + h * noise, Xu) - f(ell, scale, noise - h * noise, Xu)) / (2 * h * noise)
        g_xu = np.empty_like(Xu)
        for idx in np.ndindex(*Xu.shape):
            a, c = Xu.copy(), Xu.copy()
            a[idx] += h
            c[idx] -= h
            g_xu[idx] = (f(ell, scale, noise, a) - f(ell, scale, noise, c)) / (2 * h)
        K = ref.get_kernel(name)
        p = {"k_length": ell, "k_scale": scale, "noise": noise}
        Kuu = K(Xu, Xu, p, jitter=jitter)
        Kuf = K(Xu, self.X, p)
        Q = Kuf.T @ np.linalg.solve(Kuu, Kuf) + noise * np.eye(self.N)
        dy = -np.linalg.solve(Q, yres)
        return b, 0, dict(k_length=g_ell, k_scale=g_s, noise=g_n, Xu=g_xu, yres=dy)

    def sgp_posterior(self, kind, ell, scale, noise, jitter, Xu, yres, Xnew, noise_p, want_cov=True, want_var=False):
        p = {"k_length": broadcast_lengthscale(ell, self.d), "k_scale": scale, "noise": noise}
        noiseless = noise_p == 0.0 and noise != 0.0
        mean, cov = ref.sparse_posterior(self.X, np.asarray(yres, dtype=np.float64), np.asarray(Xu, dtype=np.float64),
                                         np.asarray(Xnew, dtype=np.float64), p, noiseless, kernel=_NAMES[kind],
                                         jitter=jitter)
        mean = np.atleast_1d(mean)
        return mean, (cov if want_cov else None), (np.diag(cov).copy() if want_var else None), 0
