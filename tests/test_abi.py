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


def test_chain_kernels_fit_beside_two_resident_trailing_update_workgroups():
    """What makes the diagonal-block kernel of the blocked Cholesky (gpax/models/gp.py:160-164) cheap inside the pipeline
    is that it is PLACED AT ONCE: the trailing SYRK keeps two workgroups resident per CU, and what they leave free is
    512 - 2 x alloc(trailing) VGPRs per SIMD lane and 160 KB - 2 x 64 KB of LDS.  Read from the code objects hipcc emits
    for the sources as committed (no GPU needed): the default potf2 kernel and the latency GEMM shapes of the panel
    chain fit in that, the round-3 kernel (kept as the reference) does not."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import kernel_resources as kr

    csrc = os.path.join(os.path.dirname(__file__), "..", "gpax_amd", "csrc")
    files = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    assert {"potf2.hip", "gemm_f64.hip", "fit_small.hip", "linalg.hip", "sparse.hip", "gram.hip", "api.hip"} <= set(files)
    rows = {r["name"]: r for f in files for r in kr.resources(os.path.join(csrc, f))}

    def find(sub):
        hit = [r for n, r in rows.items() if sub in n]
        assert len(hit) == 1, (sub, [n for n in rows if sub in n])
        return hit[0]

    def alloc(r):  # VGPRs are allocated in blocks of 8 on gfx90a+
        return (int(r["vgpr_count"]) + 7) // 8 * 8

    trailing = find("gemm_nt128_kernelILi1ELi1E")  # TAG = 1 (Cholesky trailing update), accumulators from -C
    free_vgpr = 512 - 2 * alloc(trailing)
    free_lds = 160 * 1024 - 2 * (2 * 256 * 16 * 8)
    assert free_vgpr >= 104 and free_lds == 32 * 1024
    slim = find("potf2_slim_kernel")
    assert alloc(slim) <= free_vgpr, (slim["vgpr_count"], free_vgpr)
    assert int(slim["vgpr_spill_count"]) == 0 and int(slim["private_segment_fixed_size"]) == 0
    assert (13 * 16 * 17 + 64) * 8 <= free_lds  # POTF2_SLIM_LDS (dynamic LDS: csrc/potf2_slim.h)
    assert not [n for n in rows if "potf2_chain_kernel" in n]  # round 3's 344-VGPR kernel left the product (tools/exp/)
    # round-1 register-staged latency shapes (64x64, 32x128 strips): the chain launches of every BLOCKED factorisation
    # (csrc/linalg.hip lat_now = 1: C3, C4, every N > 5120) — BK = 16 fits beside two trailing-update workgroups in every
    # epilogue form (EPI 0 / 1 / 2 are kernels of their own since round 6)
    r1 = [r for n, r in rows.items() if "gemm_nt_kernel" in n]
    assert len(r1) == 12  # 2 shapes x BK 16 / 32 x 3 epilogue forms
    for r in r1:
        if "Li16ELb0E" in r["name"]:
            assert alloc(r) <= free_vgpr, (r["name"], r["vgpr_count"])
    # round 5 latency shapes (gemm_tile.h lat_tile), every epilogue form: <= 80 VGPRs, no scratch at all, and their
    # LDS rings — 3 x (64 + 64) x 64 B and 2 x (32 + 128) x 64 B — fit in the 28 KB the diagonal-block kernel is known to
    # be placed with
    lat = [r for n, r in rows.items() if "gemm_lat_kernel" in n]
    assert len(lat) == 6
    for r in lat:
        assert alloc(r) <= 80 and alloc(r) <= free_vgpr, (r["name"], r["vgpr_count"])
        assert int(r["vgpr_spill_count"]) == 0 and int(r["sgpr_spill_count"]) == 0 and int(r["private_segment_fixed_size"]) == 0
    assert 3 * 128 * 64 <= 28 * 1024 and 2 * 160 * 64 <= 28 * 1024
    assert len([n for n in rows if "fit_small_kernel" in n]) == 15  # 3 kernels x (d = 1 .. 4, generic d)
    # the fused chain step (potf2 + panel TRSM strips in one launch, round 6): within its 128-VGPR cap, no SGPR spill either
    fused = find("potf2_trsm_kernel")
    assert int(fused["vgpr_count"]) <= 128 and int(fused["sgpr_spill_count"]) == 0
    # the persistent big-tile GEMM re-reads its argument block per tile instead of holding it in SGPRs across the tile loop
    # (33 - 39 of them went to VGPR lanes until round 6)
    persist = [r for n, r in rows.items() if "gemm_nt128_persist_kernel" in n]
    assert len(persist) == 6 and all(int(r["sgpr_spill_count"]) == 0 for r in persist)
    # no VGPR spill and no private segment in ANY kernel of the library (every .hip under csrc/)
    assert len(rows) > 100
    for n, r in rows.items():
        assert int(r["vgpr_spill_count"]) == 0 and int(r["private_segment_fixed_size"]) == 0, n


def test_the_fused_chain_step_publishes_with_a_release_and_its_strips_do_not_invalidate_the_l2():
    """potf2.hip potf2_trsm_kernel, read from the code object hipcc emits (no GPU needed): workgroup 0 ends with ONE
    agent-scope release (`buffer_wbl2 sc1`, then the flag store `sc1`); the strip part — everything before the first
    `s_endpgm` — polls the flag with an L2-bypassing load (`sc1`) between `s_sleep`s, carries NO `buffer_inv` (the per-wave
    acquire made every strip workgroup re-fetch L^-1 from memory: profiles/r06/potf2_trsm.md; the comment in the kernel says
    why none is needed), brings its 32 rows of L^-1 in with 32 LDS-direct loads, passes ONE barrier (before the flag) and
    issues 2 accumulators x 32 k-steps of v_mfma_f64_16x16x4_f64 (gp.py:160-164's Cholesky, the panel TRSM of a step)."""
    import os
    import re
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import kernel_resources as kr

    asm = kr.kernel_asm(os.path.join(os.path.dirname(__file__), "..", "gpax_amd", "csrc", "potf2.hip"), "potf2_trsm_kernel")
    ends = [m.start() for m in re.finditer(r"s_endpgm", asm)]
    assert len(ends) == 2
    strips, wg0 = asm[:ends[0]], asm[ends[0]:]
    assert "buffer_inv" not in asm
    assert len(re.findall(r"buffer_wbl2 sc1", wg0)) == 1 and "buffer_wbl2" not in strips
    assert re.search(r"global_store_dword [^\n]* sc1", wg0[wg0.index("buffer_wbl2 sc1"):])
    assert "s_sleep" in strips and re.search(r"global_load_dword [^\n]* sc1", strips)
    assert len(re.findall(r"buffer_load_dwordx4 [^\n]* lds", strips)) == 32
    assert len(re.findall(r"v_mfma_f64_16x16x4", strips)) == 64
    assert len(re.findall(r"s_barrier", strips)) == 1
    assert strips.index("s_barrier") < strips.index("s_sleep")
