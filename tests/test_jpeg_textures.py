"""JPEG image maps (map_Ka / map_Kd / map_Ks; the reference decodes them with SDL_image, i.e.
libjpeg, obj_reader.c:352-381).  The loader's decoder follows libjpeg's default stages (accurate
integer IDCT, fancy chroma upsampling, fixed-point colour tables); the fixtures hold the pixels
libjpeg-turbo itself (through Pillow) decoded from each file
(tests/golden/make_jpeg_fixtures.py), and the loader must reproduce them exactly."""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

from raytracer_amd import _lib
from test_png_textures import _load_with_maps, _texture

JPG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg")
EXPECTED = np.load(os.path.join(JPG, "expected_rgb.npz"))


@pytest.mark.parametrize("name", sorted(EXPECTED.files))
def test_jpeg_decodes_to_libjpegs_pixels(tmp_path, name):
    shutil.copy(os.path.join(JPG, name + ".jpg"), tmp_path / (name + ".jpg"))
    lib, h = _load_with_maps(tmp_path, f"map_Ka {name}.jpg\n")
    assert lib.rt_obj_num_textures(h) == 1, "the file was not decoded"
    got = _texture(lib, h, 0)
    want = EXPECTED[name]
    assert got.shape == want.shape
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() == 0, f"{int((d > 0).sum())} samples differ, by up to {d.max()}"
    lib.rt_obj_free(h)


@pytest.mark.parametrize("name", ["refused_progressive.jpg", "refused_truncated.jpg"])
def test_unsupported_jpegs_behave_like_a_failed_img_load(tmp_path, name):
    shutil.copy(os.path.join(JPG, name), tmp_path / name)
    lib, h = _load_with_maps(tmp_path, f"map_Kd {name}\n")
    v = _lib.RtObjView()
    assert lib.rt_obj_get(h, C.byref(v)) == 0
    assert lib.rt_obj_num_textures(h) == 0 and v.map_Kd[0] == -1
    lib.rt_obj_free(h)
