"""Host-side logic and the C ABI surface, without a GPU."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import _lib, kernel, parallel, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n.startswith("rt_") or n in
                      ("mat_mult_device", "intersect_all_triangles_device")))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rt_amd.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    assert lib.rt_device_count() == 0
    assert lib.rt_init(0) == -5                      # RT_E_NODEVICE
    assert b"no HIP device" in lib.rt_last_error()
    h = C.c_void_p()
    assert lib.rt_scene_create(C.byref(h)) == -5     # nothing falls back to the CPU
    assert lib.rt_multi_create(C.byref(h), 2, None) == -5 and not h.value
    assert lib.rt_multi_num_devices(None) == 0 and lib.rt_multi_destroy(None) == 0
    assert lib.rt_multi_exchange(None) == b""


def test_reference_struct_layout():
    # sizeof(Triangle) = 488, mtl at 40, Ka/Kd/Ks at 304/328/352 (obj_reader.h:7-39, LP64)
    mesh = scenes.with_materials(scenes.torus(40), num_mats=2, unassigned_every=3)
    ta = kernel.TriangleArray(mesh)
    assert ta.stride == 488
    raw = ta.buf.tobytes()
    for i in (0, 1, 5):
        a, b, c = struct.unpack_from("<3i", raw, 488 * i)
        assert (a, b, c) == (mesh.A[i], mesh.B[i], mesh.C[i])
        assert struct.unpack_from("<i", raw, 488 * i + 40)[0] == mesh.mtl[i]
        if mesh.mtl[i] >= 0:
            assert struct.unpack_from("<3d", raw, 488 * i + 328) == tuple(mesh.Kd[mesh.mtl[i]])


def test_band_partition():
    for H in (1, 7, 480, 1080, 2160):
        for n in (1, 2, 3, 4, 8):
            prev = 0
            for r in range(n):
                y0, y1 = kernel.band(H, n, r)
                assert y0 == prev and y1 >= y0
                prev = y1
            assert prev == H
            sizes = [kernel.band(H, n, r)[1] - kernel.band(H, n, r)[0] for r in range(n)]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == parallel.max_band_rows(H, n)
    lib = _lib.load()
    a, b = C.c_int(), C.c_int()
    assert lib.rt_band(10, 0, 0, C.byref(a), C.byref(b)) != 0
    assert lib.rt_band(10, 2, 2, C.byref(a), C.byref(b)) != 0


def test_placement_matches_oracle():
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(9, 5, 3), dtype=np.uint8)
    assert np.array_equal(parallel.place_rows_host(rgb), O.place_pixels(rgb))


def test_bmp_writer(tmp_path):
    # graphic_tools.c:19-23,115-121: 32 bpp, masks R=00FF0000 G=0000FF00 B=000000FF A=FF000000
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(6, 4, 3), dtype=np.uint8)
    p = str(tmp_path / "pic0000.bmp")
    kernel.save_bmp(p, img)
    raw = open(p, "rb").read()
    assert raw[:2] == b"BM"
    off = struct.unpack_from("<I", raw, 10)[0]
    w, h = struct.unpack_from("<ii", raw, 18)
    bpp = struct.unpack_from("<H", raw, 28)[0]
    masks = struct.unpack_from("<4I", raw, 54)
    assert (w, h, bpp) == (4, 6, 32)
    assert masks == (0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000)
    px = np.frombuffer(raw, dtype=np.uint8, offset=off).reshape(6, 4, 4)[::-1]   # bottom-up rows
    assert np.array_equal(px[..., 2], img[..., 0]) and np.array_equal(px[..., 1], img[..., 1])
    assert np.array_equal(px[..., 0], img[..., 2]) and np.all(px[..., 3] == 255)
    try:
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), img)
    except ImportError:
        pass
    assert _lib.load().rt_save_bmp(b"/nonexistent_dir/x.bmp", img.ctypes.data, 4, 6) == -4


def test_scene_generators_are_seeded_and_well_formed():
    a, b = scenes.triangle_soup(500), scenes.triangle_soup(500)
    assert np.array_equal(a.x, b.x) and np.array_equal(a.A, b.A)
    for m in (a, scenes.torus(300), scenes.sphere(300), scenes.torus(200, with_uv=True)):
        for idx in (m.A, m.B, m.C):
            assert idx.min() >= 1 and idx.max() <= m.num_v
        for idx in (m.An, m.Bn, m.Cn):
            assert idx.min() >= 1 and idx.max() <= m.num_vn
        ext = max(np.ptp(m.x[1:]), np.ptp(m.y[1:]), np.ptp(m.z[1:]))
        assert ext == pytest.approx(15.0, rel=1e-9)


def test_torus_quads_split_like_the_reference():
    # obj_reader.c:167-231: quad (a,b,c,d) -> (a,b,c) then (a,c,d)
    m = scenes.torus(64)
    assert m.num_tris % 2 == 0
    assert np.array_equal(m.A[0::2], m.A[1::2]) and np.array_equal(m.C[0::2], m.B[1::2])
