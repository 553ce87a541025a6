"""Launch shape of the primary sweep (rt_raster_shape through rt_raster_shape_query): host logic
only, so it runs without a GPU (the library then assumes 256 compute units).

The wave-block form (blocks of 4096 x 64 NL pixels, a workgroup each) is the default wherever
blocks x triangle chunks give two rounds of workgroups; smaller meshes keep the lattice form, in
which the pixels supply the workgroups; rows that are not regular keep the lattice form with a
coordinate per row.  RT_AMD_RASTER_FORCE overrides the choice for the tests of every shape."""
import ctypes as C

import pytest

from raytracer_amd import _lib

BLOCK_SHAPES = [4096 * nl for nl in (1, 2, 4, 8, 16, 36)]
TILE = 128


def shape(W, rows, tris, regular=True):
    lib = _lib.load()
    ppl, chunks, wx, wgs = C.c_int(), C.c_int(), C.c_int(), C.c_long()
    rc = lib.rt_raster_shape_query(W, rows, tris, 1 if regular else 0, C.byref(ppl), C.byref(chunks), C.byref(wx),
                                   C.byref(wgs))
    assert rc == 0
    return ppl.value, chunks.value, wx.value, wgs.value


@pytest.fixture(autouse=True)
def no_overrides(monkeypatch):
    for k in ("RT_AMD_RASTER_FORCE", "RT_AMD_CULL", "RT_AMD_RASTER_MAX_CHUNKS", "RT_AMD_RASTER_ROUNDS"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("W,rows,tris", [(3840, 2160, 1_000_000), (3840, 270, 1_000_000), (3840, 2160, 100_000),
                                         (3840, 2160, 69_451), (1920, 1080, 100_000), (7680, 4320, 2_000_000)])
def test_large_meshes_take_the_wave_block_form(W, rows, tris):
    ppl, chunks, wx, wgs = shape(W, rows, tris)
    tiles = (tris + TILE - 1) // TILE
    assert ppl in BLOCK_SHAPES and wx == -1
    block_rows = ppl // 64
    assert wgs == -(-W // 4096) * -(-rows // block_rows)            # a workgroup per block
    assert 1 <= chunks <= min(tiles, 4096)                          # (twice the cap the shapes are compared under)
    assert wgs * min(tiles, 2048) >= 2 * 8 * 256                    # two rounds of the machine were available
    # no shape with fewer instructions per triangle over all blocks was passed over
    cost = wgs * (98 + 1.5 * (ppl // 4096))
    for other in BLOCK_SHAPES:
        o_wgs = -(-W // 4096) * -(-rows // (other // 64))
        if o_wgs * min(tiles, 2048) >= 2 * 8 * 256:
            assert cost <= o_wgs * (98 + 1.5 * (other // 4096)) + 1e-9


def test_the_headline_frame():
    # 1 M triangles at 3840 x 2160: three blocks of 4096 x 1024 (one block of 2304 rows would leave a
    # single round of workgroups at the chunk cap)
    assert shape(3840, 2160, 1_000_000) == (65536, 4096, -1, 3)


@pytest.mark.parametrize("W,rows,tris", [(3840, 2160, 6_320), (640, 480, 6_320), (160, 120, 500), (37, 19, 700)])
def test_small_meshes_keep_the_lattice_form(W, rows, tris):
    ppl, chunks, wx, wgs = shape(W, rows, tris)
    tiles = (tris + TILE - 1) // TILE
    assert 1 <= ppl <= 1024 and ppl & (ppl - 1) == 0 and wx == -1    # regular rows: a workgroup per wave-sized block
    assert 1 <= chunks <= tiles and wgs >= 1


@pytest.mark.parametrize("tris", [500, 100_000, 1_000_000])
def test_rows_that_are_not_regular_keep_a_coordinate_per_row(tris):
    ppl, chunks, wx, wgs = shape(3840, 2160, tris, regular=False)
    assert ppl <= 2048 and wx == 0 and chunks >= 1


def test_tile_culling_keeps_narrow_lattices(monkeypatch):
    monkeypatch.setenv("RT_AMD_CULL", "1")
    lib = _lib.load()
    if lib.rt_get_tile_culling() == 0:
        pytest.skip("the culling switch was set by a call in this process; the variable no longer decides")
    ppl, chunks, wx, wgs = shape(3840, 2160, 1_000_000)
    assert ppl <= 32 and wx == 0


@pytest.mark.parametrize("force,want", [("147456,3,-1", (147456, 3, -1)), ("65536,5", (65536, 5, -1)),
                                        ("4096,2,1", (4096, 2, -1)),          # block shapes always own a block each
                                        ("2048,2,1", (2048, 2, 1)), ("16,3,4", (16, 3, 4)), ("256,2", (256, 2, -1))])
def test_forced_shapes(force, want, monkeypatch):
    monkeypatch.setenv("RT_AMD_RASTER_FORCE", force)
    ppl, chunks, wx, _ = shape(500, 300, 700)
    assert (ppl, chunks, wx) == want


@pytest.mark.parametrize("force", ["3000,2", "12288,2", "4096,0", "4096,7", "banana"])
def test_invalid_forced_shapes_are_ignored(force, monkeypatch):
    free = shape(500, 300, 700)                 # 6 tiles: "4096,7" asks for more chunks than tiles
    monkeypatch.setenv("RT_AMD_RASTER_FORCE", force)
    assert shape(500, 300, 700) == free


def test_block_shapes_need_regular_rows(monkeypatch):
    monkeypatch.setenv("RT_AMD_RASTER_FORCE", "8192,2,-1")
    ppl, _, _, _ = shape(500, 300, 700, regular=False)
    assert ppl <= 2048
