"""Row sets (rt_rows): interleaved parts cover every row exactly once, in ascending order, and
the contiguous band is the parts = 1 case.  Host-only: no GPU needed."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import _lib, kernel, parallel


@pytest.mark.parametrize("H,parts,block", [(2160, 8, 16), (480, 3, 16), (37, 4, 5), (16, 8, 16), (1, 1, 16),
                                           (100, 7, 1), (2160, 1, 16), (5, 8, 16), (21, 8, 16), (1080, 8, 16)])
def test_parts_partition_the_image(H, parts, block):
    seen = []
    for p in range(parts):
        r = kernel.rows_part(H, parts, p, block)
        rows = kernel.rows_image_rows(r)
        assert len(rows) == kernel.rows_count(r)
        assert np.all(np.diff(rows) > 0)
        lib = _lib.load()
        for lr in (0, len(rows) // 2, len(rows) - 1):
            if len(rows):
                assert lib.rt_rows_image_row(C.byref(r), lr) == rows[lr]
        seen.append(rows)
    allrows = np.sort(np.concatenate(seen))
    assert np.array_equal(allrows, np.arange(H))
    assert parallel.max_part_rows(H, parts, block) == max(len(s) for s in seen)
    # the parts differ by one row at most
    assert max(len(s) for s in seen) - min(len(s) for s in seen) <= 1


def test_blocks_are_dealt_after_the_thue_morse_sequence():
    # 3 parts, blocks of 2 rows: rounds 0 1 2 3 are dealt forwards, backwards, backwards, forwards
    got = [list(kernel.rows_image_rows(kernel.rows_part(24, 3, p, 2))) for p in range(3)]
    assert got[0] == [0, 1, 10, 11, 16, 17, 18, 19]
    assert got[1] == [2, 3, 8, 9, 14, 15, 20, 21]
    assert got[2] == [4, 5, 6, 7, 12, 13, 22, 23]
    # the rows after the last complete round are shared out evenly, in that round's direction
    got = [list(kernel.rows_image_rows(kernel.rows_part(9, 3, p, 2))) for p in range(3)]
    assert got == [[0, 1, 8], [2, 3, 7], [4, 5, 6]]
    got = [list(kernel.rows_image_rows(kernel.rows_part(11, 3, p, 2))) for p in range(3)]
    assert got == [[0, 1, 9, 10], [2, 3, 7, 8], [4, 5, 6]]
    # 4K over 8 parts: 16 complete rounds and 14 of the last 112 rows each
    assert [kernel.rows_count(kernel.rows_part(2160, 8, p, 16)) for p in range(8)] == [270] * 8


def test_sub_range_and_band():
    r = kernel.rows_part(100, 3, 1, 8, y0=10, y1=75)
    rows = kernel.rows_image_rows(r)
    assert rows[0] == 18 and rows.max() < 75 and rows.min() >= 10
    band = kernel.rows_part(100, 1, 0, 25, y0=50, y1=75)
    assert np.array_equal(kernel.rows_image_rows(band), np.arange(50, 75))


@pytest.mark.parametrize("bad", [dict(block=0), dict(parts=0), dict(part=3, parts=3), dict(part=-1)])
def test_bad_row_sets_are_rejected(bad):
    kw = dict(parts=2, part=0, block=16)
    kw.update(bad)
    r = kernel.rows_part(64, kw["parts"], kw["part"], kw["block"])
    assert _lib.load().rt_rows_count(C.byref(r)) < 0
    assert b"row set" in _lib.load().rt_last_error()


def test_rt_band_gives_the_first_ranks_the_extra_rows():
    """rt_band: contiguous bands of H // n rows, the first H % n ranks one row more (a C caller
    sizes its buffers from this rule)."""
    for H, n in ((2160, 8), (1080, 7), (10, 3), (5, 8), (480, 1)):
        bands = [kernel.band(H, n, r) for r in range(n)]
        assert bands[0][0] == 0 and bands[-1][1] == H
        assert all(bands[r][1] == bands[r + 1][0] for r in range(n - 1))
        sizes = [y1 - y0 for y0, y1 in bands]
        assert sizes == [H // n + (1 if r < H % n else 0) for r in range(n)]
