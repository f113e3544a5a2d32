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
