"""PNG image maps (map_Ka / map_Kd / map_Ks; the reference decodes them with SDL_image,
obj_reader.c:352-381).  The loader's decoder against fixtures whose pixels are known from how
they were made (tests/golden/make_png_fixtures.py: an encoder written from the PNG specification,
cross-checked with Pillow there), host side; then a textured quad through loader, scene and
shading against the oracle fed with the same pixels, on the GPU."""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

from oracle import obj_oracle, oracle as O
from raytracer_amd import _lib, kernel, scenes
from conftest import TAN_HALF

PNG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png")
EXPECTED = np.load(os.path.join(PNG, "expected_rgb.npz"))

QUAD = ("mtllib m.mtl\nv -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nvt 0.02 0.03\nvt 0.97 0.05\nvt 0.95 0.98\nvt 0.04 0.96\n"
        "vn 0 0 1\nusemtl tex\nf 1/1/1 2/2/1 3/3/1 4/4/1\n")


def _load_with_maps(tmp_path, lines):
    (tmp_path / "m.mtl").write_text("newmtl tex\nKa 0.3 0.4 0.5\nKd 0.6 0.5 0.4\nKs 0.2 0.2 0.2\n" + lines)
    (tmp_path / "m.obj").write_text(QUAD)
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.rt_obj_load(str(tmp_path / "m.obj").encode(), C.byref(h)), "rt_obj_load")
    return lib, h


def _texture(lib, h, k):
    w, hh, px = C.c_int(), C.c_int(), C.c_void_p()
    assert lib.rt_obj_texture(h, k, C.byref(w), C.byref(hh), C.byref(px)) == 0
    buf = (C.c_uint8 * (w.value * hh.value * 3)).from_address(px.value)
    return np.frombuffer(buf, np.uint8).reshape(hh.value, w.value, 3).copy()


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_png_decodes_to_the_known_pixels(tmp_path, name):
    shutil.copy(os.path.join(PNG, name + ".png"), tmp_path / (name + ".png"))
    lib, h = _load_with_maps(tmp_path, f"map_Kd {name}.png\n")
    v = _lib.RtObjView()
    assert lib.rt_obj_get(h, C.byref(v)) == 0
    assert lib.rt_obj_num_textures(h) == 1 and v.map_Kd[0] == 0 and v.map_Ka[0] == -1
    got = _texture(lib, h, 0)
    assert got.shape == EXPECTED[name].shape and np.array_equal(got, EXPECTED[name])
    lib.rt_obj_free(h)


@pytest.mark.parametrize("name", ["bad_truncated.png", "bad_signature.png", "missing.png", "photo.jpg"])
def test_undecodable_maps_behave_like_a_failed_img_load(tmp_path, name):
    if name.startswith("bad"):
        shutil.copy(os.path.join(PNG, name), tmp_path / name)
    elif name == "photo.jpg":
        (tmp_path / name).write_bytes(b"\xff\xd8\xff\xe0" + bytes(64))       # a JPEG header: no decoder for it
    lib, h = _load_with_maps(tmp_path, f"map_Kd {name}\n")
    v = _lib.RtObjView()
    assert lib.rt_obj_get(h, C.byref(v)) == 0
    assert lib.rt_obj_num_textures(h) == 0 and v.map_Kd[0] == -1             # NULL map: constant Kd is used
    assert lib.rt_obj_texture(h, 0, None, None, None) == -1
    lib.rt_obj_free(h)


def _png_chunk(kind, body):
    import struct
    import zlib
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def _huge_header_files():
    """Small files whose headers declare enormous pictures: each must be refused before anything
    is allocated for it (a texture file is untrusted input)."""
    import struct
    import zlib
    png = (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", 32768, 32768, 16, 6, 0, 0, 0)) +
           _png_chunk(b"IDAT", zlib.compress(bytes(64))) + _png_chunk(b"IEND", b""))
    # the same size within the pixel cap's reach but far beyond what 80 bytes of IDAT can inflate to
    png2 = (b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", 8000, 8000, 8, 2, 0, 0, 0)) +
            _png_chunk(b"IDAT", zlib.compress(bytes(64))) + _png_chunk(b"IEND", b""))
    # baseline JPEG: SOI, a quantisation table, SOF0 declaring 65535 x 65535 YCbCr 4:2:0, Huffman
    # tables, SOS and a few bytes of scan data
    dqt = b"\xff\xdb" + struct.pack(">H", 67) + b"\x00" + bytes([1] * 64)
    sof = b"\xff\xc0" + struct.pack(">HBHHB", 17, 8, 65535, 65535, 3) + bytes([1, 0x22, 0, 2, 0x11, 0, 3, 0x11, 0])
    def dht(tc_th):
        counts = [0] * 16
        counts[0] = 2
        return b"\xff\xc4" + struct.pack(">H", 2 + 1 + 16 + 2) + bytes([tc_th]) + bytes(counts) + bytes([0, 1])
    sos = b"\xff\xda" + struct.pack(">HB", 12, 3) + bytes([1, 0, 2, 0, 3, 0]) + bytes([0, 63, 0])
    jpg = b"\xff\xd8" + dqt + sof + dht(0x00) + dht(0x10) + sos + bytes(40) + b"\xff\xd9"
    jpg_small = jpg.replace(struct.pack(">HH", 65535, 65535), struct.pack(">HH", 8000, 8000))
    ppm = b"P6\n100000 100000\n255\n" + bytes(30)
    bmp = bytearray(54 + 16)
    bmp[0:2] = b"BM"
    struct.pack_into("<I", bmp, 10, 54)
    struct.pack_into("<iiHH", bmp, 18, 60000, 60000, 1, 24)
    return {"huge.png": png, "huge2.png": png2, "huge.jpg": jpg, "huge2.jpg": jpg_small, "huge.ppm": ppm,
            "huge.bmp": bytes(bmp)}


@pytest.mark.parametrize("name", sorted(_huge_header_files()))
def test_huge_headers_are_refused_without_allocating(tmp_path, name):
    import resource
    (tmp_path / name).write_bytes(_huge_header_files()[name])
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    lib, h = _load_with_maps(tmp_path, f"map_Kd {name}\n")
    v = _lib.RtObjView()
    assert lib.rt_obj_get(h, C.byref(v)) == 0
    assert lib.rt_obj_num_textures(h) == 0 and v.map_Kd[0] == -1             # like a failed IMG_Load
    lib.rt_obj_free(h)
    # (ru_maxrss is in KiB: nothing like the gigabytes the headers ask for was ever touched)
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - before < 512 * 1024


@pytest.mark.gpu
def test_png_textured_quad_against_the_oracle(tmp_path):
    import torch
    for n in ("rgb8_32x16", "pal4_adam7", "grey8"):
        shutil.copy(os.path.join(PNG, n + ".png"), tmp_path / (n + ".png"))
    lib, h = _load_with_maps(tmp_path, "map_Kd rgb8_32x16.png\nmap_Ka pal4_adam7.png\nmap_Ks grey8.png\n")
    assert lib.rt_obj_num_textures(h) == 3
    # the oracle: its own parser for the geometry, the fixtures' known pixels as textures
    ref = obj_oracle.load(str(tmp_path / "m.obj"))
    mesh = scenes.Mesh(**{k: np.ascontiguousarray(ref[k]) for k in
                          ("x", "y", "z", "xn", "yn", "zn", "u", "v", "A", "B", "C", "An", "Bn", "Cn", "At", "Bt", "Ct")},
                       mtl=np.ascontiguousarray(ref["mtl"]), Ka=ref["Ka"], Kd=ref["Kd"], Ks=ref["Ks"],
                       map_Kd=np.array([0], np.int32), map_Ka=np.array([1], np.int32), map_Ks=np.array([2], np.int32),
                       textures=[EXPECTED["rgb8_32x16"], EXPECTED["pal4_adam7"], EXPECTED["grey8"]])
    fr = kernel.driver_frame(3)
    W, H = 96, 72
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    want = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF)
    sc = kernel.Scene()
    _lib.check(lib.rt_scene_set_obj(sc.h, h), "rt_scene_set_obj")
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF)
    assert (want["hit"] >= 0).sum() > 300
    assert np.array_equal(out["hit"].cpu().numpy(), want["hit"])
    assert np.abs(out["argb"].cpu().numpy() - want["argb"]).max() < 1e-9
    assert np.abs(out["rgb8"].cpu().numpy().astype(int) - want["rgb8"].astype(int)).max() <= 1
    # the picture really shows the texture: many distinct colours on one flat quad
    m = want["hit"] >= 0
    assert len(np.unique(want["rgb8"][m], axis=0)) > 30
    lib.rt_obj_free(h)
    sc.close()
