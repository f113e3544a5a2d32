"""CPU: the C-ABI shared library loads and exports every symbol include/gpx.h declares; the
product fails loudly without a GPU (no fallback)."""
import os
import re

import pytest

from gpax_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load_library()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gpx.h but not exported by libgpx.so"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names


def test_no_cpu_fallback_engine_raises_without_gpu():
    import ctypes
    lib = _lib.load_library()
    ctx = ctypes.c_void_p()
    rc = lib.gpx_init(0, ctypes.byref(ctx))
    if rc == 0:  # a GPU is present (this file also runs on the GPU box): nothing to assert
        lib.gpx_destroy(ctx)
        pytest.skip("GPU present")
    assert rc < 0 and b"HIP" in lib.gpx_last_error(ctx) or b"device" in lib.gpx_last_error(ctx)
    lib.gpx_destroy(ctx)
    with pytest.raises(_lib.GpxError):
        _lib.Engine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gpax_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("oracle/", "").replace("the oracle", "") or "import" not in [
                    l for l in src.splitlines() if "oracle" in l and "import" in l] or False, f
                for line in src.splitlines():
                    assert not re.match(r"\s*(from|import)\s+oracle", line), (f, line)


def test_xcd_aware_tile_order_covers_every_tile_once_and_balances_the_xcds():
    """The XCD-aware order of the big-tile GEMM (gemm_f64.hip make_swz_map / swz_decode, run on the host through the
    diagnostic entry point): every tile of the launch exactly once — lower launches only those on or below the
    diagonal, for square, near (ti_off > tj_off) and rectangular shapes — and the same tile count per XCD to within a
    couple of items of 8 tiles."""
    import ctypes as C
    import numpy as np
    from gpax_amd import _lib

    lib = _lib.load_library()
    for lower, ti, tj, tm, tn in [(1, 0, 0, 77, 77), (1, 5, 5, 40, 40), (1, 12, 4, 30, 38), (1, 4, 4, 124, 124),
                                  (0, 0, 0, 33, 8), (0, 3, 0, 16, 64), (1, 0, 0, 9, 9), (1, 20, 0, 50, 20)]:
        cap = tm * tn
        out = np.zeros(3 * cap, dtype=np.int32)
        n = lib.gpx_debug_tile_order(lower, ti, tj, tm, tn, out.ctypes.data_as(C.POINTER(C.c_int)), cap)
        want = {(by, bx) for by in range(tm) for bx in range(tn) if not lower or tj + bx <= ti + by}
        assert n == len(want), (lower, ti, tj, tm, tn, n, len(want))
        trip = out[:3 * n].reshape(n, 3)
        got = [(int(b), int(c)) for _, b, c in trip]
        assert len(set(got)) == n and set(got) == want
        per_xcd = np.bincount(trip[:, 0], minlength=8)
        if n >= 1024:
            assert per_xcd.max() - per_xcd.min() <= 24, per_xcd
    assert lib.gpx_debug_tile_order(1, 0, 0, 8 * 40, 8, None, 0) == -1  # more strips than the map holds: grid order
