"""The tile lists of the plain GEMM launches (csrc/gemm_f64.hip plain_launch_map / decode_tile; round 5): since a lower-tile
launch runs over a LIST of its live tiles instead of a square grid, a wrong decode is a tile computed twice or never — the
Cholesky under gpax/models/gp.py:160-164 with a hole in its trailing update.  Host only: the library decodes the list with
the code the kernels run, the test enumerates by brute force."""
import numpy as np
import pytest

from gpax_amd import _lib


def _live(delta, tm, tn):
    return [(by, bx) for by in range(tm) for bx in range(tn) if bx <= by + delta]


@pytest.mark.parametrize("delta", [-5, -1, 0, 1, 2, 7, 40])
@pytest.mark.parametrize("tm,tn", [(1, 1), (2, 2), (3, 7), (7, 3), (16, 16), (31, 31), (62, 62), (80, 64), (17, 250), (256, 256)])
def test_lower_lists_hold_every_live_tile_once_row_by_row(delta, tm, tn):
    got = [tuple(t) for t in _lib.tile_list(True, delta, tm, tn)]
    assert got == _live(delta, tm, tn)  # row-major: a row-trimmed launch (k range starts at the row tile) starts its longest rows first


@pytest.mark.parametrize("tm,tn", [(1, 1), (4, 4), (5, 9), (32, 32), (64, 16)])
def test_full_lists_in_the_three_orders(tm, tn):
    rows = [tuple(t) for t in _lib.tile_list(False, 0, tm, tn, 0)]
    assert rows == [(by, bx) for by in range(tm) for bx in range(tn)]
    last_first = [tuple(t) for t in _lib.tile_list(False, 0, tm, tn, 1)]
    assert last_first == [(by, bx) for bx in reversed(range(tn)) for by in range(tm)]  # kupper: the longest column first
    first_first = [tuple(t) for t in _lib.tile_list(False, 0, tm, tn, 2)]
    assert first_first == [(by, bx) for bx in range(tn) for by in range(tm)]  # kcol


def test_large_lower_list_and_bad_arguments():
    t = _lib.tile_list(True, 0, 1290, 1290)  # the 64 x 64 tiles of N = 82 560: indices beyond 2^19, where a float sqrt would slip
    assert len(t) == 1290 * 1291 // 2 and tuple(t[-1]) == (1289, 1289) and tuple(t[0]) == (0, 0)
    by = t[:, 0].astype(np.int64)
    assert np.array_equal(np.bincount(by), np.arange(1, 1291)) and np.all(np.diff(by) >= 0)
    assert np.all(t[:, 1] <= t[:, 0])
    assert len(_lib.tile_list(True, -9, 4, 4)) == 0
    with pytest.raises(ValueError):
        _lib.tile_list(False, 0, 4, 4, 3)
