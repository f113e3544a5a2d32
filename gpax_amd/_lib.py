"""
ctypes binding of libgpx (include/gpx.h) and the `Engine` object the model classes drive.

There is NO CPU fallback in this package: if the shared library is missing or no gfx950 device
is present, constructing an Engine raises.  (tests/ inject a checker-backed engine to exercise the
host-side samplers on a CPU-only machine; the product never does.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

KERNEL_KINDS = {"RBF": 0, "Matern": 1, "Periodic": 2}
KIND_R2 = 3  # gpx_gram only (include/gpx.h GPX_KERNEL_R2): the squared scaled distance itself
KIND_PERIODIC = 2
MAX_DIM = 16

PROF_GEMM_TRAILING, PROF_GEMM_OTHER, PROF_POTF2, PROF_GRAM = 0, 1, 2, 3
STAGE_GRAM, STAGE_POTRF, STAGE_FITSTEP, STAGE_POSTERIOR, STAGE_PREDICT = 0, 1, 2, 3, 4

# GPX_LIB: another build of the same ABI (e.g. the AddressSanitizer build, `make -C gpax_amd/csrc asan`)
_LIB_PATH = os.environ.get("GPX_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libgpx.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class GpxError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load_library() -> C.CDLL:
    """dlopen libgpx.so (built in-tree by `__graft_entry__.build()` / gpax_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise GpxError(
            f"libgpx.so not found at {_LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gpax_amd/csrc`). gpax_amd has no CPU fallback."
        )
    lib = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    sig = {
        "gpx_init": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "gpx_device_count": (C.c_int, []),
        "gpx_device_pci": (C.c_int, [C.c_int, _ip, _ip, _ip]),
        "gpx_destroy": (None, [vp]),
        "gpx_last_error": (C.c_char_p, [vp]),
        "gpx_device_info": (C.c_int, [vp, C.c_char_p, C.c_int, _ip, C.POINTER(C.c_int64), _ip]),
        "gpx_synchronize": (C.c_int, [vp]),
        "gpx_gram": (C.c_int, [vp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_double,
                               C.c_double, C.c_int, _dp]),
        "gpx_set_train": (C.c_int, [vp, _dp, C.c_int, C.c_int]),
        "gpx_factor": (C.c_int, [vp, C.c_int, _dp, C.c_double, C.c_double, C.c_double, _dp, _dp, _ip]),
        "gpx_lml_grad": (C.c_int, [vp, _dp, _dp, _dp, _dp]),
        "gpx_fit_batch": (C.c_int, [vp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp, C.c_int, _dp, _ip, _dp,
                                    _dp]),
        "gpx_set_train_tasks": (C.c_int, [vp, _dp, C.c_int, C.c_int, C.c_int]),
        "gpx_posterior": (C.c_int, [vp, _dp, C.c_int, C.c_double, C.c_double, _dp, _dp, _dp]),
        "gpx_mvn_draw": (C.c_int, [vp, _dp, C.c_int, _dp, _ip]),
        "gpx_predict_sweep": (C.c_int, [vp, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp,
                                        C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, _dp, _ip, _dp, _dp, C.c_int]),
        "gpx_lml_grad_diag": (C.c_int, [vp, _dp]),
        "gpx_set_diag": (C.c_int, [vp, _dp, C.c_int]),
        "gpx_sgp_bound": (C.c_int, [vp, C.c_int, _dp, C.c_double, C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_int,
                                    _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
        "gpx_sgp_posterior": (C.c_int, [vp, C.c_int, _dp, C.c_double, C.c_double, C.c_double, _dp, C.c_int, _dp, _dp,
                                        C.c_int, C.c_double, _dp, _dp, _dp, _ip]),
        "gpx_profile_enable": (C.c_int, [vp, C.c_int]),
        "gpx_profile_reset": (C.c_int, [vp]),
        "gpx_profile_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int64), _dp, _dp]),
        "gpx_profile_read_bytes": (C.c_int, [vp, C.c_int, _dp]),
        "gpx_debug_set_potf2": (C.c_int, [vp, C.c_char_p]),
        "gpx_debug_set_lat_gemm": (C.c_int, [vp, C.c_char_p]),
        "gpx_nuts_transition": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, _ip, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp,
                                          _dp, _dp, _dp, _dp, C.c_double, _dp, C.c_int, C.POINTER(C.c_uint64), _dp, _ip, _ip]),
        "gpx_debug_pcg64_doubles": (C.c_int, [C.POINTER(C.c_uint64), C.c_int, _dp]),
        "gpx_nuts_potential": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, _ip, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp,
                                         _dp, _dp, _dp]),
        "gpx_debug_set_serialise_trailing": (C.c_int, [vp, C.c_int]),
        "gpx_debug_gemm_time": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
        "gpx_debug_tile_list": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip]),
        "gpx_time_stage": (C.c_int, [vp, C.c_int, C.c_int, _dp]),
        "gpx_sweep_resident": (C.c_int, [vp, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, _dp]),
        "gpx_sweep_stats": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _ip]),
        "gpx_mfma_f64_peak": (C.c_int, [vp, _dp]),
        "gpx_gemm_nt": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_double, _dp]),
        "gpx_potrf": (C.c_int, [vp, C.c_int, _dp, _dp, _ip]),
        "gpx_node_init": (C.c_int, [C.c_int, _ip, C.c_int, C.POINTER(vp)]),
        "gpx_node_destroy": (None, [vp]),
        "gpx_node_last_error": (C.c_char_p, [vp]),
        "gpx_node_info": (C.c_int, [vp, _ip, _ip, _ip, _ip]),
        "gpx_node_last_shares": (C.c_int, [vp, _ip, C.c_int]),
        "gpx_predict_sweep_multi": (C.c_int, [vp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int,
                                              _dp, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, _dp, _ip, _dp,
                                              C.c_int]),
        "gpx_rank_unique_id": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
        "gpx_rank_init": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]),
        "gpx_rank_destroy": (None, [vp]),
        "gpx_rank_last_error": (C.c_char_p, [vp]),
        "gpx_rank_info": (C.c_int, [vp, _ip, _ip, _ip, _ip, _ip]),
        "gpx_rank_device_pci": (C.c_int, [vp, _ip, _ip, _ip]),
        "gpx_rank_collective_calls": (C.c_int64, [vp]),
        "gpx_rank_barrier": (C.c_int, [vp]),
        "gpx_rank_calibrate": (C.c_int, [vp, _dp]),
        "gpx_shard_ranges_weighted": (C.c_int, [C.c_int, _dp, C.c_int, _ip, _ip]),
        "gpx_rank_allreduce_max": (C.c_int, [vp, _dp, C.c_int]),
        "gpx_rank_bcast": (C.c_int, [vp, _dp, C.c_int64]),
        "gpx_rank_predict_sweep": (C.c_int, [vp, C.c_int, _dp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int,
                                             _dp, C.c_int, C.c_int, C.c_double, _dp, C.c_int, _dp, _dp, _ip, _dp,
                                             C.c_int, C.c_int]),
        "gpx_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, _ip, _ip]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "gpx_init gpx_device_count gpx_device_pci gpx_destroy gpx_last_error gpx_device_info gpx_synchronize gpx_gram gpx_set_train gpx_set_train_tasks gpx_set_diag "
    "gpx_factor gpx_lml_grad gpx_lml_grad_diag gpx_fit_batch gpx_posterior gpx_mvn_draw gpx_predict_sweep gpx_sgp_bound gpx_sgp_posterior "
    "gpx_profile_enable "
    "gpx_profile_reset gpx_profile_read gpx_profile_read_bytes gpx_debug_set_potf2 gpx_debug_set_lat_gemm gpx_nuts_transition gpx_nuts_potential gpx_debug_pcg64_doubles gpx_debug_set_serialise_trailing gpx_debug_gemm_time gpx_debug_tile_list gpx_time_stage gpx_sweep_resident gpx_sweep_stats gpx_mfma_f64_peak gpx_gemm_nt "
    "gpx_potrf gpx_node_init gpx_node_destroy gpx_node_last_error gpx_node_info gpx_node_last_shares gpx_predict_sweep_multi "
    "gpx_rank_unique_id gpx_rank_init gpx_rank_destroy gpx_rank_last_error gpx_rank_info gpx_rank_device_pci gpx_rank_collective_calls gpx_rank_barrier gpx_rank_calibrate gpx_shard_ranges_weighted "
    "gpx_rank_allreduce_max gpx_rank_bcast gpx_rank_predict_sweep gpx_shard_range"
).split()


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_dp)


def kernel_kind(name) -> int:
    try:
        return KERNEL_KINDS[name]
    except (KeyError, TypeError):
        raise NotImplementedError(
            f"kernel {name!r} has no MI355X path; the HIP Gram kernels cover {sorted(KERNEL_KINDS)}") from None


def broadcast_lengthscale(k_length, d: int) -> np.ndarray:
    """k_length may be scalar, (d,), (1,), (1,1), (1,d) — gpax/tests/test_kernels.py:19,38,
    gpax/tests/test_vigp.py:73-75."""
    ell = np.asarray(k_length, dtype=np.float64).reshape(-1)
    if ell.size == 1:
        ell = np.full(d, ell[0])
    if ell.size != d:
        raise ValueError(f"k_length has {ell.size} entries for input_dim {d}")
    return np.ascontiguousarray(ell)


def pack_ell(kind: int, k_length, d: int, period=None) -> np.ndarray:
    """The `ell` argument of the C-ABI: d lengthscales, followed by the period for the periodic
    kernel (include/gpx.h GPX_KERNEL_PERIODIC).  An already packed (d + 1,) array passes through."""
    if kind != KIND_PERIODIC:
        return broadcast_lengthscale(k_length, d)
    arr = np.asarray(k_length, dtype=np.float64).reshape(-1)
    if period is None:
        if arr.size != d + 1:
            raise ValueError("the periodic kernel needs `period` (or a packed k_length of d + 1 values)")
        return np.ascontiguousarray(arr)
    per = float(np.asarray(period, dtype=np.float64).reshape(-1)[0])
    return np.ascontiguousarray(np.concatenate([broadcast_lengthscale(arr, d), [per]]))


def n_ell(kind: int, d: int) -> int:
    return d + (1 if kind == KIND_PERIODIC else 0)


class Engine:
    """One libgpx context = one GPU.  Thin, stateful mirror of the C-ABI."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.gpx_init(int(device), C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.gpx_last_error(self._ctx).decode() if self._ctx else "gpx_init failed"
            if self._ctx:
                self._lib.gpx_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise GpxError(f"gpx_init(device={device}) failed: {msg}. gpax_amd needs an MI355X (gfx950); "
                           "there is no CPU fallback.")
        self.device = int(device)
        self.N = 0
        self.d = 0
        self.M = 0
        self.T = 1
        self._diag_key = None
        self._train_owner = None
        self._train_version = None

    # -- plumbing -----------------------------------------------------------------------------
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise GpxError(f"{what} failed ({rc}): {self._lib.gpx_last_error(self._ctx).decode()}")

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.gpx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cu, hbm, clk = C.c_int(), C.c_int64(), C.c_int()
        self._check(self._lib.gpx_device_info(self._ctx, name, 256, C.byref(cu), C.byref(hbm), C.byref(clk)),
                    "gpx_device_info")
        return {"name": name.value.decode(), "num_cu": cu.value, "hbm_bytes": hbm.value, "clock_khz": clk.value}

    def synchronize(self):
        self._check(self._lib.gpx_synchronize(self._ctx), "gpx_synchronize")

    # -- Gram ---------------------------------------------------------------------------------
    def gram(self, kind: int, X, Z, ell, scale: float, diag_add: float, add_diag: bool) -> np.ndarray:
        X = _f64(X)
        Z = _f64(Z)
        if X.ndim != 2 or Z.ndim != 2 or X.shape[1] != Z.shape[1]:
            raise ValueError(f"gram: X {X.shape} and Z {Z.shape} must be (n, d) and (m, d)")
        n, d = X.shape
        m = Z.shape[0]
        ell = pack_ell(kind, ell, d)
        out = np.empty((n, m), dtype=np.float64)
        self._check(self._lib.gpx_gram(self._ctx, kind, _ptr(X), n, _ptr(Z), m, d, _ptr(ell), float(scale),
                                       float(diag_add), int(bool(add_diag)), _ptr(out)), "gpx_gram")
        return out

    # -- exact GP -----------------------------------------------------------------------------
    def set_train(self, X):
        """X (N, d), or (T, N, d): T task-specific training sets (vExactGP) — then the batched entry points
        treat entry b as task b % T."""
        X = _f64(X)
        self._diag_key = None  # gpx_set_train clears the per-point diagonal
        # whoever calls set_train replaces the resident training set: no model may go on believing it owns it
        # (the owner re-registers itself right after its own set_train, models/gp.py:_engine)
        self._train_owner = None
        self._train_version = None
        if X.ndim == 3:
            self.T, self.N, self.d = X.shape
            self._check(self._lib.gpx_set_train_tasks(self._ctx, _ptr(X), self.T, self.N, self.d),
                        "gpx_set_train_tasks")
            return
        self.T = 1
        self.N, self.d = X.shape
        self._check(self._lib.gpx_set_train(self._ctx, _ptr(X), self.N, self.d), "gpx_set_train")

    def factor(self, kind: int, ell, scale: float, noise: float, jitter: float, yres) -> Tuple[float, int]:
        ell = pack_ell(kind, ell, self.d)
        self._last_kind = kind
        yres = _f64(yres, (self.N,))
        lml, info = C.c_double(), C.c_int()
        self._check(self._lib.gpx_factor(self._ctx, kind, _ptr(ell), float(scale), float(noise), float(jitter),
                                         _ptr(yres), C.byref(lml), C.byref(info)), "gpx_factor")
        return lml.value, info.value

    def lml_grad(self):
        g_ell = np.empty(n_ell(getattr(self, '_last_kind', 0), self.d))
        g_scale, g_noise = C.c_double(), C.c_double()
        alpha = np.empty(self.N)
        self._check(self._lib.gpx_lml_grad(self._ctx, _ptr(g_ell), C.byref(g_scale), C.byref(g_noise),
                                           _ptr(alpha)), "gpx_lml_grad")
        return g_ell, g_scale.value, g_noise.value, alpha

    def lml_grad_diag(self) -> np.ndarray:
        """d lml / d v of the per-point diagonal set by set_diag; call after lml_grad()."""
        g = np.empty(self.N)
        self._check(self._lib.gpx_lml_grad_diag(self._ctx, _ptr(g)), "gpx_lml_grad_diag")
        return g

    def fit_batch(self, kind: int, ells, scales, noises, jitter: float, yres, want_grad: bool = True):
        """gpx_factor + gpx_lml_grad for B hyper-parameter vectors in one launch sequence.
        ells (B, n_ell); yres (N,) shared or (B, N).  Returns lml (B,), info (B,), grad (B, n_ell + 2) or None
        ([d/d ell.., d/d scale, d/d noise]), alpha (B, N) or None."""
        ells = _f64(ells)
        B = ells.shape[0]
        ne = n_ell(kind, self.d)
        ells = _f64(ells, (B, ne))
        scales, noises = _f64(scales, (B,)), _f64(noises, (B,))
        yres = _f64(yres)  # (N,) shared, (B, N) per entry or (T, N) per task
        rows = 1 if yres.ndim == 1 else yres.shape[0]
        yres = _f64(yres, (rows, self.N))
        lml = np.empty(B)
        info = np.zeros(B, dtype=np.int32)
        grad = np.empty((B, ne + 2)) if want_grad else None
        alpha = np.empty((B, self.N)) if want_grad else None
        self._last_kind = kind
        self._check(self._lib.gpx_fit_batch(self._ctx, kind, B, _ptr(ells), _ptr(scales), _ptr(noises), float(jitter),
                                            _ptr(yres), rows, _ptr(lml), info.ctypes.data_as(_ip), _ptr(grad),
                                            _ptr(alpha)), "gpx_fit_batch")
        return lml, info, grad, alpha

    def nuts_plan(self, kind: int, idx_ell, idx_scale: int, idx_noise: int, loc, scale, const, jitter: float, yres):
        """The constant arguments of gpx_nuts_transition, converted once per chain (see nuts_transition)."""
        idx = np.ascontiguousarray(idx_ell, dtype=np.int32)
        return dict(kind=int(kind), dim=int(idx.size) + 2, ne=int(idx.size), idx=idx, idx_scale=int(idx_scale),
                    idx_noise=int(idx_noise), loc=_f64(loc).copy(), scale=_f64(scale).copy(), const=_f64(const).copy(),
                    jitter=float(jitter), yres=_f64(yres, (self.N,)).copy(), state=(C.c_uint64 * 4)(), acc=C.c_double(),
                    nl=C.c_int(), div=C.c_int(), U=C.c_double())

    def nuts_potential(self, plan: dict, u):
        """(U, g) of the potential gpx_nuts_transition integrates, at u (gpx_nuts_potential)."""
        u = np.ascontiguousarray(u, dtype=np.float64)
        g = np.empty_like(u)
        U = C.c_double()
        self._check(self._lib.gpx_nuts_potential(
            self._ctx, plan["kind"], plan["dim"], plan["ne"], plan["idx"].ctypes.data_as(_ip), plan["idx_scale"],
            plan["idx_noise"], _ptr(plan["loc"]), _ptr(plan["scale"]), _ptr(plan["const"]), plan["jitter"], _ptr(plan["yres"]),
            _ptr(u), C.byref(U), _ptr(g)), "gpx_nuts_potential")
        return U.value, g

    def nuts_transition(self, plan: dict, u, U: float, g, p0, eps: float, inv_mass, max_tree_depth: int, rng):
        """One NUTS transition in the library (gpx_nuts_transition, csrc/nuts.hip): (u, U, g, accept, n_leapfrog,
        diverging) — the uniforms come out of `rng` (a Generator over PCG64), which is advanced exactly as the Python
        loop of gpax_amd/infer/nuts.py would advance it."""
        bg = rng.bit_generator
        st = bg.state
        s_, i_ = st["state"]["state"], st["state"]["inc"]
        state = plan["state"]
        state[0], state[1], state[2], state[3] = s_ >> 64, s_ & 0xFFFFFFFFFFFFFFFF, i_ >> 64, i_ & 0xFFFFFFFFFFFFFFFF
        u = np.array(u, dtype=np.float64)
        g = np.array(g, dtype=np.float64)
        p0 = np.ascontiguousarray(p0, dtype=np.float64)
        im = np.ascontiguousarray(inv_mass, dtype=np.float64)
        plan["U"].value = float(U)
        self._check(self._lib.gpx_nuts_transition(
            self._ctx, plan["kind"], plan["dim"], plan["ne"], plan["idx"].ctypes.data_as(_ip), plan["idx_scale"],
            plan["idx_noise"], _ptr(plan["loc"]), _ptr(plan["scale"]), _ptr(plan["const"]), plan["jitter"], _ptr(plan["yres"]),
            _ptr(u), C.byref(plan["U"]), _ptr(g), _ptr(p0), float(eps), _ptr(im), int(max_tree_depth), state,
            C.byref(plan["acc"]), C.byref(plan["nl"]), C.byref(plan["div"])), "gpx_nuts_transition")
        st["state"]["state"] = (int(state[0]) << 64) | int(state[1])
        bg.state = st
        self._last_kind = plan["kind"]
        return u, plan["U"].value, g, plan["acc"].value, plan["nl"].value, bool(plan["div"].value)

    def _xu(self, Xu) -> np.ndarray:
        Xu = _f64(Xu)
        if Xu.ndim != 2 or Xu.shape[1] != self.d:
            raise ValueError(f"inducing points have shape {Xu.shape}; expected (M_ind, d={self.d})")
        return Xu

    def _xnew(self, Xnew) -> np.ndarray:
        """Test inputs as the C-ABI reads them: (M, d) — or (T, M, d) with task-specific training sets."""
        Xnew = _f64(Xnew)
        want = 3 if getattr(self, "T", 1) > 1 else 2
        if Xnew.ndim != want or Xnew.shape[-1] != self.d or (want == 3 and Xnew.shape[0] != self.T):
            raise ValueError(f"X_new has shape {Xnew.shape}; expected "
                             f"{'(T=%d, M, d=%d)' % (self.T, self.d) if want == 3 else '(M, d=%d)' % self.d}")
        return Xnew

    def posterior(self, Xnew, noise_p: float, jitter: float, want_cov: bool = True, want_var: bool = False):
        Xnew = self._xnew(Xnew)
        M = Xnew.shape[0]
        self.M = M
        mean = np.empty(M)
        cov = np.empty((M, M)) if want_cov else None
        var = np.empty(M) if want_var else None
        self._check(self._lib.gpx_posterior(self._ctx, _ptr(Xnew), M, float(noise_p), float(jitter), _ptr(mean),
                                            _ptr(cov), _ptr(var)), "gpx_posterior")
        return mean, cov, var

    def mvn_draw(self, eps) -> Tuple[np.ndarray, int]:
        eps = _f64(eps)
        n = eps.shape[0]
        out = np.empty((n, self.M))
        info = C.c_int()
        self._check(self._lib.gpx_mvn_draw(self._ctx, _ptr(eps), n, _ptr(out), C.byref(info)), "gpx_mvn_draw")
        return out, info.value

    def set_diag(self, v) -> None:
        """Per-point variances added to K's diagonal (None clears them); see gpx_set_diag."""
        if v is None:
            self._check(self._lib.gpx_set_diag(self._ctx, None, 0), "gpx_set_diag")
            return
        v = _f64(v, (self.N,))
        self._check(self._lib.gpx_set_diag(self._ctx, _ptr(v), self.N), "gpx_set_diag")

    def predict_sweep(self, kind: int, ells, scales, noises, yres, Xnew, noiseless: bool, jitter: float,
                      eps: Optional[np.ndarray], want_var: bool = False, pred_diag=None, m_slice: int = 0):
        ells = _f64(ells)
        S = ells.shape[0]
        ells = _f64(ells, (S, n_ell(kind, self.d)))
        scales = _f64(scales, (S,))
        noises = _f64(noises, (S,))
        yres = _f64(yres)  # (N,) shared, (S, N) per sample or (T, N) per task
        rows = 1 if yres.ndim == 1 else yres.shape[0]
        yres = _f64(yres, (rows, self.N))
        Xnew = self._xnew(Xnew)  # (M, d), or (T, M, d) after set_train with task-specific inputs
        M = Xnew.shape[-2]
        self.M = M
        n = 0 if eps is None else int(np.asarray(eps).shape[1])
        eps_c = None if n == 0 else _f64(eps, (S, n, M))
        means = np.empty((S, M))
        samples = np.empty((S, n, M))
        infos = np.zeros(S, dtype=np.int32)
        vars_ = np.empty((S, M)) if want_var else None
        pd = None if pred_diag is None else _f64(pred_diag, (S, M))
        self._check(self._lib.gpx_predict_sweep(
            self._ctx, kind, S, _ptr(ells), _ptr(scales), _ptr(noises), _ptr(yres), rows, _ptr(Xnew), M,
            int(bool(noiseless)), float(jitter), _ptr(eps_c), n, _ptr(means),
            _ptr(samples) if n else None, infos.ctypes.data_as(_ip), _ptr(vars_), _ptr(pd), int(m_slice)),
            "gpx_predict_sweep")
        if want_var:
            return means, samples, infos, vars_
        return means, samples, infos

    # -- sparse GP ------------------------------------------------------------------------------
    def sgp_bound(self, kind: int, ell, scale: float, noise: float, jitter: float, Xu, yres, want_grad: bool = True):
        """VFE bound of viSparseGP.model and (optionally) its gradient w.r.t. (ell, scale, noise, Xu)
        plus d bound / d yres.  Returns (bound, info, grads-or-None)."""
        ell = broadcast_lengthscale(ell, self.d)
        Xu = self._xu(Xu)
        Mi = Xu.shape[0]
        yres = _f64(yres, (self.N,))
        bound, info = C.c_double(), C.c_int()
        if want_grad:
            g_ell, g_Xu, dy = np.empty(self.d), np.empty((Mi, self.d)), np.empty(self.N)
            g_scale, g_noise = C.c_double(), C.c_double()
            self._check(self._lib.gpx_sgp_bound(self._ctx, kind, _ptr(ell), float(scale), float(noise), float(jitter),
                                                _ptr(Xu), Mi, _ptr(yres), 1, C.byref(bound), _ptr(g_ell),
                                                C.byref(g_scale), C.byref(g_noise), _ptr(g_Xu), _ptr(dy),
                                                C.byref(info)), "gpx_sgp_bound")
            return bound.value, info.value, dict(k_length=g_ell, k_scale=g_scale.value, noise=g_noise.value, Xu=g_Xu,
                                                 yres=dy)
        self._check(self._lib.gpx_sgp_bound(self._ctx, kind, _ptr(ell), float(scale), float(noise), float(jitter),
                                            _ptr(Xu), Mi, _ptr(yres), 0, C.byref(bound), None, None, None, None, None,
                                            C.byref(info)), "gpx_sgp_bound")
        return bound.value, info.value, None

    def sgp_posterior(self, kind: int, ell, scale: float, noise: float, jitter: float, Xu, yres, Xnew, noise_p: float,
                      want_cov: bool = True, want_var: bool = False):
        ell = broadcast_lengthscale(ell, self.d)
        Xu = self._xu(Xu)
        yres = _f64(yres, (self.N,))
        Xnew = self._xnew(Xnew)
        Ms = Xnew.shape[0]
        mean = np.empty(Ms)
        cov = np.empty((Ms, Ms)) if want_cov else None
        var = np.empty(Ms) if want_var else None
        info = C.c_int()
        self._check(self._lib.gpx_sgp_posterior(self._ctx, kind, _ptr(ell), float(scale), float(noise), float(jitter),
                                                _ptr(Xu), Xu.shape[0], _ptr(yres), _ptr(Xnew), Ms, float(noise_p),
                                                _ptr(mean), _ptr(cov), _ptr(var), C.byref(info)), "gpx_sgp_posterior")
        return mean, cov, var, info.value

    # -- measurement ----------------------------------------------------------------------------
    def profile_enable(self, on: bool):
        self._check(self._lib.gpx_profile_enable(self._ctx, int(on)), "gpx_profile_enable")

    def profile_reset(self):
        self._check(self._lib.gpx_profile_reset(self._ctx), "gpx_profile_reset")

    def profile_read(self, cls: int):
        n, ms, work = C.c_int64(), C.c_double(), C.c_double()
        self._check(self._lib.gpx_profile_read(self._ctx, cls, C.byref(n), C.byref(ms), C.byref(work)),
                    "gpx_profile_read")
        return n.value, ms.value, work.value

    def profile_read_bytes(self, cls: int) -> float:
        b = C.c_double()
        self._check(self._lib.gpx_profile_read_bytes(self._ctx, cls, C.byref(b)), "gpx_profile_read_bytes")
        return b.value

    def set_potf2(self, mode: str) -> None:
        """Diagonal-block kernel of the blocked Cholesky for the following calls: 'slim' (default) or 'tile' —
        the same bits from both (gpx_debug_set_potf2; bench.py's in-process A/B)."""
        self._check(self._lib.gpx_debug_set_potf2(self._ctx, mode.encode()), "gpx_debug_set_potf2")

    def set_lat_gemm(self, mode: str) -> None:
        """Latency-shape GEMM of the panel chains for the following calls: 'auto' (default), 'r5' or 'r1' — same bits."""
        self._check(self._lib.gpx_debug_set_lat_gemm(self._ctx, mode.encode()), "gpx_debug_set_lat_gemm")

    def set_serialise_trailing(self, on: bool) -> None:
        """Measurement mode: the Cholesky trailing updates of the following calls run alone on the chip (gpx.h)."""
        self._check(self._lib.gpx_debug_set_serialise_trailing(self._ctx, int(bool(on))), "gpx_debug_set_serialise_trailing")

    def gemm_time(self, tiles_m: int, tiles_n: int, K: int, mode: int = 1, lower: bool = False, shape: int = 0,
                  reps: int = 20) -> float:
        """Milliseconds per launch of one library GEMM on resident scratch operands (gpx_debug_gemm_time)."""
        ms = C.c_double()
        self._check(self._lib.gpx_debug_gemm_time(self._ctx, int(tiles_m), int(tiles_n), int(K), int(mode), int(bool(lower)),
                                                  int(shape), int(reps), C.byref(ms)), "gpx_debug_gemm_time")
        return ms.value

    def time_stage(self, stage: int, reps: int) -> float:
        ms = C.c_double()
        self._check(self._lib.gpx_time_stage(self._ctx, stage, reps, C.byref(ms)), "gpx_time_stage")
        return ms.value

    def sweep_resident(self, kind: int, ells, scales, noises, noiseless: bool, jitter: float, n_draws: int) -> float:
        ells = _f64(ells)
        S = ells.shape[0]
        ells = _f64(ells, (S, n_ell(kind, self.d)))
        scales = _f64(scales, (S,))
        noises = _f64(noises, (S,))
        ms = C.c_double()
        self._check(self._lib.gpx_sweep_resident(self._ctx, kind, S, _ptr(ells), _ptr(scales), _ptr(noises),
                                                 int(bool(noiseless)), float(jitter), int(n_draws), C.byref(ms)),
                    "gpx_sweep_resident")
        return ms.value

    def sweep_stats(self):
        """(batches launched, samples processed, batch size B of the last sweep) of this context."""
        nb, ns, last = C.c_int64(0), C.c_int64(0), C.c_int(0)
        self._check(self._lib.gpx_sweep_stats(self._ctx, C.byref(nb), C.byref(ns), C.byref(last)), "gpx_sweep_stats")
        return int(nb.value), int(ns.value), int(last.value)

    def mfma_f64_peak(self) -> float:
        return self.mfma_f64_probe()["tflops"]

    def mfma_f64_probe(self) -> dict:
        out = (C.c_double * 3)()
        self._check(self._lib.gpx_mfma_f64_peak(self._ctx, out), "gpx_mfma_f64_peak")
        return {"tflops": out[0], "cycles_per_mfma": out[1], "effective_mhz": out[2]}

    # -- unit-test entry points -------------------------------------------------------------------
    def gemm_nt(self, A, B, alpha=1.0, beta=0.0, Cin=None) -> np.ndarray:
        A = _f64(A)
        B = _f64(B)
        M, K = A.shape
        N = B.shape[0]
        Cm = np.zeros((M, N)) if Cin is None else _f64(Cin).copy()
        self._check(self._lib.gpx_gemm_nt(self._ctx, M, N, K, float(alpha), _ptr(A), _ptr(B), float(beta),
                                          _ptr(Cm)), "gpx_gemm_nt")
        return Cm

    def potrf(self, A) -> Tuple[np.ndarray, int]:
        A = _f64(A)
        n = A.shape[0]
        L = np.empty((n, n))
        info = C.c_int()
        self._check(self._lib.gpx_potrf(self._ctx, n, _ptr(A), _ptr(L), C.byref(info)), "gpx_potrf")
        return L, info.value


class Node:
    """The GPUs of one node behind ONE process (include/gpx.h gpx_node_*): `inflight` libgpx contexts on each of
    `devices` plus one RCCL communicator per device.  predict_sweep = Engine.predict_sweep with the posterior
    samples split in contiguous blocks over the GPUs: ncclBroadcast of the inputs, ncclSend / ncclRecv gather of the
    results, over xGMI — the vmap axis of ExactGP.predict (gpax/models/gp.py:392-395), SURVEY.md 8e."""

    def __init__(self, devices=None, inflight: Optional[int] = None):
        self._lib = load_library()
        self._node = C.c_void_p()
        if devices is None:
            devices = list(range(visible_device_count()))
        elif isinstance(devices, int):
            devices = list(range(devices))
        self.devices = [int(v) for v in devices]
        if not self.devices:
            raise GpxError("Node: no GPU visible")
        self.inflight = sweep_inflight() if inflight is None else max(1, int(inflight))
        arr = (C.c_int * len(self.devices))(*self.devices)
        rc = self._lib.gpx_node_init(len(self.devices), arr, self.inflight, C.byref(self._node))
        if rc != 0:
            msg = self._lib.gpx_node_last_error(self._node).decode() if self._node else "gpx_node_init failed"
            if self._node:
                self._lib.gpx_node_destroy(self._node)
                self._node = C.c_void_p()
            raise GpxError(f"gpx_node_init(devices={self.devices}) failed ({rc}): {msg}")

    def close(self):
        if getattr(self, "_node", None):
            self._lib.gpx_node_destroy(self._node)
            self._node = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        g, f, t, v = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = self._lib.gpx_node_info(self._node, C.byref(g), C.byref(f), C.byref(t), C.byref(v))
        if rc != 0:
            raise GpxError("gpx_node_info failed")
        return {"ngpu": g.value, "inflight": f.value, "transport": "rccl" if t.value else "memcpy",
                "rccl_version": v.value}

    def last_shares(self) -> list:
        """Samples every GPU worked off in the last predict_sweep (the deal is dynamic: gpx_node_last_shares)."""
        arr = (C.c_int * len(self.devices))()
        g = self._lib.gpx_node_last_shares(self._node, arr, len(self.devices))
        return [int(arr[i]) for i in range(max(g, 0))]

    def predict_sweep(self, X, kind: int, ells, scales, noises, yres, Xnew, noiseless: bool, jitter: float,
                      eps: Optional[np.ndarray], want_var: bool = False, m_slice: int = 0):
        X = _f64(X)
        N, d = X.shape
        ells = _f64(ells)
        S = ells.shape[0]
        ells = _f64(ells, (S, n_ell(kind, d)))
        scales, noises = _f64(scales, (S,)), _f64(noises, (S,))
        yres = _f64(yres)
        rows = 1 if yres.ndim == 1 else yres.shape[0]
        yres = _f64(yres, (rows, N))
        Xnew = _f64(Xnew)
        if Xnew.ndim != 2 or Xnew.shape[1] != d:
            raise ValueError(f"X_new has shape {Xnew.shape}; expected (M, d={d})")
        M = Xnew.shape[0]
        n = 0 if eps is None else int(np.asarray(eps).shape[1])
        eps_c = None if n == 0 else _f64(eps, (S, n, M))
        means = np.empty((S, M))
        samples = np.empty((S, n, M))
        infos = np.zeros(S, dtype=np.int32)
        vars_ = np.empty((S, M)) if want_var else None
        rc = self._lib.gpx_predict_sweep_multi(
            self._node, kind, _ptr(X), N, d, S, _ptr(ells), _ptr(scales), _ptr(noises), _ptr(yres), rows, _ptr(Xnew), M,
            int(bool(noiseless)), float(jitter), _ptr(eps_c), n, _ptr(means), _ptr(samples) if n else None,
            infos.ctypes.data_as(_ip), _ptr(vars_), int(m_slice))
        if rc != 0:
            raise GpxError(f"gpx_predict_sweep_multi failed ({rc}): {self._lib.gpx_node_last_error(self._node).decode()}")
        if want_var:
            return means, samples, infos, vars_
        return means, samples, infos


UNIQUE_ID_BYTES = 128


def pcg64_doubles(rng: np.random.Generator, n: int) -> np.ndarray:
    """n uniforms of `rng` (a Generator over PCG64) drawn by the library's own PCG64 (gpx_debug_pcg64_doubles) WITHOUT
    advancing `rng` — what tests compare with rng.uniform()."""
    st = rng.bit_generator.state
    s_, i_ = st["state"]["state"], st["state"]["inc"]
    state = (C.c_uint64 * 4)(s_ >> 64, s_ & 0xFFFFFFFFFFFFFFFF, i_ >> 64, i_ & 0xFFFFFFFFFFFFFFFF)
    out = np.empty(int(n))
    if load_library().gpx_debug_pcg64_doubles(state, int(n), _ptr(out)) != 0:
        raise ValueError("gpx_debug_pcg64_doubles")
    return out


def shard_ranges_weighted(S: int, weights) -> list:
    """All blocks [(lo, hi), ...] over S samples with sizes in proportion to `weights` (gpx_shard_ranges_weighted; host
    only, needs no GPU) — the blocks the ranks of a calibrated sweep work off."""
    w = _f64(weights).reshape(-1)
    parts = int(w.size)
    lo, hi = (C.c_int * parts)(), (C.c_int * parts)()
    if load_library().gpx_shard_ranges_weighted(int(S), _ptr(w), parts, lo, hi) != 0:
        raise ValueError(f"shard_ranges_weighted({S}, {weights})")
    return [(int(lo[r]), int(hi[r])) for r in range(parts)]


def shard_range(S: int, part: int, parts: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of part `part` out of `parts` over S samples — the library's own rule
    (gpx_shard_range; host only, needs no GPU)."""
    lo, hi = C.c_int(), C.c_int()
    if load_library().gpx_shard_range(int(S), int(part), int(parts), C.byref(lo), C.byref(hi)) != 0:
        raise ValueError(f"shard_range({S}, {part}, {parts})")
    return lo.value, hi.value


def tile_list(lower: bool, delta: int, tiles_m: int, tiles_n: int, order: int = 0) -> np.ndarray:
    """(by, bx) of every entry of the grid of a plain GEMM launch, in grid order (gpx_debug_tile_list; host only, needs
    no GPU): the live tiles of a lower-triangular launch, or the full grid row- / column-major."""
    n = load_library().gpx_debug_tile_list(int(bool(lower)), int(delta), int(tiles_m), int(tiles_n), int(order), 0, None)
    if n < 0:
        raise ValueError(f"tile_list({lower}, {delta}, {tiles_m}, {tiles_n}, {order})")
    out = np.zeros((n, 2), dtype=np.int32)
    if n:
        load_library().gpx_debug_tile_list(int(bool(lower)), int(delta), int(tiles_m), int(tiles_n), int(order), n,
                                           out.ctypes.data_as(_ip))
    return out


def rccl_unique_id() -> bytes:
    """The 128-byte id rank 0 creates for ncclCommInitRank (gpx_rank_unique_id)."""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    err = C.create_string_buffer(512)
    rc = load_library().gpx_rank_unique_id(buf, err, 512)
    if rc != 0:
        raise GpxError(f"gpx_rank_unique_id failed ({rc}): {err.value.decode(errors='replace')}")
    return buf.raw


class Rank:
    """One process = one GPU = one rank of the sharded predictive sweep (include/gpx.h gpx_rank_*): `inflight` libgpx
    contexts on `device` and rank `rank` of an RCCL communicator over `nranks` processes.  predict_sweep is collective:
    rank 0 passes the arrays and receives the results, the other ranks pass None and receive None.
    Transport: RCCL (unique_id from rank 0's rccl_unique_id()) or, with file_dir, files in a directory all ranks share
    (tests with several ranks on one GPU; fallback).  gpax/models/gp.py:392-395."""

    def __init__(self, device: int, rank: int, nranks: int, unique_id: Optional[bytes] = None,
                 file_dir: Optional[str] = None, inflight: Optional[int] = None):
        self._lib = load_library()
        self._rk = C.c_void_p()
        self.rank, self.nranks, self.device = int(rank), int(nranks), int(device)
        self.inflight = sweep_inflight() if inflight is None else max(1, int(inflight))
        if file_dir is None and (unique_id is None or len(unique_id) != UNIQUE_ID_BYTES):
            raise GpxError("Rank: the rccl transport needs rank 0's 128-byte unique id")
        rc = self._lib.gpx_rank_init(self.device, self.rank, self.nranks, unique_id,
                                     None if file_dir is None else os.fsencode(file_dir), self.inflight,
                                     C.byref(self._rk))
        if rc != 0:
            msg = self._lib.gpx_rank_last_error(self._rk).decode() if self._rk else "gpx_rank_init failed"
            if self._rk:
                self._lib.gpx_rank_destroy(self._rk)
                self._rk = C.c_void_p()
            raise GpxError(f"gpx_rank_init(device={device}, rank={rank}/{nranks}) failed ({rc}): {msg}")

    def close(self):
        if getattr(self, "_rk", None):
            self._lib.gpx_rank_destroy(self._rk)
            self._rk = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise GpxError(f"{what} failed ({rc}) on rank {self.rank}: {self._lib.gpx_rank_last_error(self._rk).decode()}")

    def info(self) -> dict:
        r, n, f, t, v = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.gpx_rank_info(self._rk, C.byref(r), C.byref(n), C.byref(f), C.byref(t), C.byref(v)),
                    "gpx_rank_info")
        return {"rank": r.value, "nranks": n.value, "inflight": f.value, "transport": "rccl" if t.value else "file",
                "rccl_version": v.value}

    def collective_calls(self) -> int:
        """RCCL collective / p2p calls issued by this rank so far (gpx_rank_collective_calls)."""
        return int(self._lib.gpx_rank_collective_calls(self._rk))

    def device_pci(self) -> int:
        """(domain << 16) | (bus << 8) | device of the GPU this rank drives (gpx_rank_device_pci)."""
        d, b, v = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.gpx_rank_device_pci(self._rk, C.byref(d), C.byref(b), C.byref(v)), "gpx_rank_device_pci")
        return (int(d.value) << 16) | (int(b.value) << 8) | int(v.value)

    def barrier(self):
        self._check(self._lib.gpx_rank_barrier(self._rk), "gpx_rank_barrier")

    def calibrate(self) -> np.ndarray:
        """Collective: every rank probes its GPU, the rates are exchanged, and the blocks of the following sweeps are sized
        in proportion (gpx_rank_calibrate).  Returns the relative speeds (mean 1, clamped to [0.85, 1.15])."""
        v = np.ones(self.nranks)
        self._check(self._lib.gpx_rank_calibrate(self._rk, _ptr(v)), "gpx_rank_calibrate")
        return v

    def allreduce_max(self, values) -> np.ndarray:
        v = _f64(values).reshape(-1).copy()
        self._check(self._lib.gpx_rank_allreduce_max(self._rk, _ptr(v), int(v.size)), "gpx_rank_allreduce_max")
        return v

    def bcast(self, arr: np.ndarray) -> np.ndarray:
        """In-place broadcast of a float64 array of the SAME size on every rank (rank 0's content wins)."""
        a = _f64(arr)
        if a is not arr:
            a = a.copy()
        self._check(self._lib.gpx_rank_bcast(self._rk, _ptr(a), int(a.size)), "gpx_rank_bcast")
        return a

    def predict_sweep(self, kind: int, N: int, d: int, S: int, M: int, n: int, noiseless: bool, jitter: float,
                      yres_rows: int = 1, X=None, ells=None, scales=None, noises=None, yres=None, Xnew=None, eps=None,
                      want_var: bool = False, m_slice: int = 0):
        """Collective.  Sizes / flags on every rank; arrays on rank 0 (ignored elsewhere).  Returns
        (means, samples, infos[, vars]) on rank 0, None on the other ranks."""
        root = self.rank == 0
        means = samples = infos = vars_ = None
        prep_error = None
        if root:
            try:
                X = _f64(X, (N, d))
                ells = _f64(ells, (S, n_ell(kind, d)))
                scales, noises = _f64(scales, (S,)), _f64(noises, (S,))
                yres = _f64(yres, (yres_rows, N))
                Xnew = _f64(Xnew, (M, d))
                eps = None if n == 0 else _f64(eps, (S, n, M))
                means = np.empty((S, M))
                samples = np.empty((S, n, M))
                infos = np.zeros(S, dtype=np.int32)
                vars_ = np.empty((S, M)) if want_var else None
            except Exception as ex:  # a bad array on the root must not strand the other ranks inside the collective:
                prep_error = ex       # enter it with null pointers — the library makes every rank leave with an error
                X = ells = scales = noises = yres = Xnew = eps = means = samples = infos = vars_ = None
                S = max(int(S), 1)
        else:
            X = ells = scales = noises = yres = Xnew = eps = None
        rc = self._lib.gpx_rank_predict_sweep(
            self._rk, int(kind), _ptr(X), int(N), int(d), int(S), _ptr(ells), _ptr(scales), _ptr(noises), _ptr(yres),
            int(yres_rows), _ptr(Xnew), int(M), int(bool(noiseless)), float(jitter), _ptr(eps), int(n), _ptr(means),
            _ptr(samples) if (root and n) else None, None if infos is None else infos.ctypes.data_as(_ip), _ptr(vars_),
            int(bool(want_var)), int(m_slice))
        if prep_error is not None:
            raise prep_error
        self._check(rc, "gpx_rank_predict_sweep")
        if not root:
            return None
        return (means, samples, infos, vars_) if want_var else (means, samples, infos)


def visible_device_count() -> int:
    """HIP devices visible to this process (gpx_device_count)."""
    return int(load_library().gpx_device_count())


def device_pci(device: int) -> int:
    """(domain << 16) | (bus << 8) | device of visible device `device` (gpx_device_pci); -1 when it cannot be read."""
    d, b, v = C.c_int(), C.c_int(), C.c_int()
    if load_library().gpx_device_pci(int(device), C.byref(d), C.byref(b), C.byref(v)) != 0:
        return -1
    return (d.value << 16) | (b.value << 8) | v.value


_default_node: Optional[Node] = None


def get_node(devices=None) -> Node:
    """Process-wide Node over `devices` (default: every visible GPU), rebuilt when the device list changes."""
    global _default_node
    want = None if devices is None else ([int(v) for v in devices] if not isinstance(devices, int)
                                         else list(range(devices)))
    if _default_node is None or (want is not None and _default_node.devices != want):
        _default_node = Node(want)
    return _default_node


_default_engine: Optional[Engine] = None


def get_engine(device: Optional[int] = None) -> Engine:
    """Process-wide engine (one GPU).  `device` defaults to $LOCAL_RANK or 0."""
    global _default_engine
    if device is None:
        device = int(os.environ.get("GPX_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if _default_engine is None or _default_engine.device != device:
        _default_engine = Engine(device)
    return _default_engine


def set_engine(engine) -> None:
    """Install an engine object (tests use this to inject a checker-backed engine on CPU)."""
    global _default_engine, _sweep_pool
    _default_engine = engine
    _sweep_pool = []


# ---- several samples in flight per GPU ---------------------------------------------------------
# The per-sample pipeline ends in a latency-bound tail (panel chain of the last outer blocks, the
# M x M chol of the draw) that leaves most of the chip idle; independent libgpx contexts on the same
# device fill those gaps with the GEMM-heavy head of another sample.  Measured on MI355X (tools/
# multi_ctx.py): C3 25.8 -> 31.0 posteriors/s with 3 contexts, C4 103 -> 147 with 4.
_sweep_pool: list = []


def sweep_inflight() -> int:
    return max(1, int(os.environ.get("GPX_INFLIGHT", "3")))


# Below this N one context's batched sweep (B samples per launch) saturates the chip; above it the
# batches are small (B <= ~30) and a few contexts in flight still overlap each other's serial tails.
_BATCHED_SINGLE_CTX_BELOW_N = 3000


def get_sweep_engines(device: Optional[int] = None, n: Optional[int] = None) -> list:
    """The default engine plus (n - 1) more contexts on the same GPU, for concurrent sweeps.  An
    injected (non-libgpx) engine is returned alone."""
    first = get_engine(device)
    if not isinstance(first, Engine):
        return [first]
    n = sweep_inflight() if n is None else max(1, int(n))
    global _sweep_pool
    _sweep_pool = [e for e in _sweep_pool if e.device == first.device]
    while len(_sweep_pool) < n - 1:
        _sweep_pool.append(Engine(first.device))
    return [first] + _sweep_pool[: n - 1]


def concurrent_sweep(engines: list, X, kind: int, ells, scales, noises, yres, Xnew, noiseless: bool, jitter: float,
                     eps, m_slice: int = 0):
    """gpx_predict_sweep over S samples, split in contiguous blocks across `engines` (one host
    thread per context; ctypes releases the GIL).  Same results as a single sweep, sample by sample."""
    import threading

    ells = np.asarray(ells, dtype=np.float64)
    S = ells.shape[0]
    n = max(1, min(len(engines), S))
    if np.shape(X)[0] < _BATCHED_SINGLE_CTX_BELOW_N:
        n = 1  # the batched sweep already fills the GPU from one context (measured: extra contexts do not help)
    if n == 1:
        engines[0].set_train(X)
        return engines[0].predict_sweep(kind, ells, scales, noises, yres, Xnew, noiseless, jitter, eps,
                                        m_slice=m_slice)
    yres = np.asarray(yres, dtype=np.float64)
    scales, noises = np.asarray(scales), np.asarray(noises)
    eps_a = None if eps is None else np.asarray(eps)
    # Guided self-scheduling: every context takes the next chunk of samples when it has finished its own — half of its
    # fair share of what is left, never less than one launch batch B (the library reports the B it chose after the
    # first chunk) — so all contexts stop within one batch of each other.  (Fixed blocks of S / n, rounds 1 - 3: the
    # context the hardware favoured finished early and the last samples ran with fewer of them in flight.)  Results do
    # not depend on the split: every sample's arithmetic is independent of its batch.
    lock = threading.Lock()
    state = {"next": 0, "min": 1}  # "min": the launch batch B once a context has reported it
    pieces: list = []
    err: list = []

    def take():
        with lock:
            lo = state["next"]
            if lo >= S:
                return None
            left = S - lo
            c = max(state["min"], -(-left // (2 * n)))
            c = -(-c // state["min"]) * state["min"]  # whole launch batches
            c = min(c, -(-S // n))  # ... but never more than the fair share: every context gets work when S >= n
            hi = min(S, lo + c)
            state["next"] = hi
            return lo, hi

    def work(i):
        try:
            e = engines[i]
            e.set_train(X)
            while not err:
                job = take()
                if job is None:
                    return
                lo, hi = job
                yr = yres if yres.ndim == 1 else yres[lo:hi]
                res = e.predict_sweep(kind, ells[lo:hi], scales[lo:hi], noises[lo:hi], yr, Xnew, noiseless, jitter,
                                      None if eps_a is None else eps_a[lo:hi], m_slice=m_slice)
                b = int(e.sweep_stats()[2])
                with lock:
                    pieces.append((lo, res))
                    if b > state["min"]:
                        state["min"] = b
        except Exception as ex:  # surface worker failures in the caller
            err.append(ex)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if err:
        raise err[0]
    out = [res for _, res in sorted(pieces, key=lambda p: p[0])]
    return (np.concatenate([o[0] for o in out]), np.concatenate([o[1] for o in out]),
            np.concatenate([o[2] for o in out]))
