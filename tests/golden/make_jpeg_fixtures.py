"""Writes the JPEG fixtures of tests/test_jpeg_textures.py and the RGB each must decode to.

The reference loads image maps with SDL_image, which decodes JPEG through libjpeg with its default
settings (accurate integer IDCT, fancy chroma upsampling).  Pillow decodes through the same library
family (libjpeg-turbo, whose routines are bit-identical to libjpeg's for these settings), so the
expected pixels here are Pillow's decode of each file: a recorded output of the reference's own
decoder library, not of the code under test.  Needs Pillow.

    python tests/golden/make_jpeg_fixtures.py
"""
import io
import os

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "jpeg")


def picture(rng, w, h):
    """A smooth picture with edges and some noise (so that all coefficient sizes occur)."""
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 5.0 + y / 9.0), 128 + 90 * np.cos(x / 3.0) * np.sin(y / 4.0),
                    (x * 7 + y * 13) % 256], 2)
    img[h // 3: h // 3 + max(h // 5, 1), w // 4: w // 4 + max(w // 3, 1)] = (250, 20, 30)
    img += rng.normal(0, 12, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(77)
    cases = [   # name, w, h, mode, save options
        ("s444_q90", 37, 29, "RGB", dict(quality=90, subsampling=0)),
        ("s422_q75", 50, 23, "RGB", dict(quality=75, subsampling=1)),
        ("s420_q85", 61, 45, "RGB", dict(quality=85, subsampling=2)),
        ("s420_q30_odd", 33, 17, "RGB", dict(quality=30, subsampling=2)),
        ("s420_optimized", 64, 48, "RGB", dict(quality=80, subsampling=2, optimize=True)),
        ("s420_16x16", 16, 16, "RGB", dict(quality=95, subsampling=2)),
        ("s422_1x1", 1, 1, "RGB", dict(quality=90, subsampling=1)),
        ("s420_3x50", 3, 50, "RGB", dict(quality=70, subsampling=2)),
        ("grey_q80", 41, 30, "L", dict(quality=80)),
        ("s420_restart", 72, 40, "RGB", dict(quality=85, subsampling=2, restart_marker_blocks=3)),
        ("s444_restart_rows", 40, 40, "RGB", dict(quality=60, subsampling=0, restart_marker_rows=1)),
        ("s420_q100", 29, 31, "RGB", dict(quality=100, subsampling=2)),
    ]
    expected = {}
    for name, w, h, mode, opts in cases:
        img = picture(rng, w, h)
        pil = Image.fromarray(img if mode == "RGB" else img[..., 0], mode)
        path = os.path.join(OUT, name + ".jpg")
        pil.save(path, "JPEG", **opts)
        expected[name] = np.asarray(Image.open(path).convert("RGB")).copy()
    # a progressive file: refused by the loader (the map then behaves like a failed IMG_Load)
    Image.fromarray(picture(rng, 24, 24), "RGB").save(os.path.join(OUT, "refused_progressive.jpg"), "JPEG",
                                                      quality=80, progressive=True)
    good = open(os.path.join(OUT, "s420_q85.jpg"), "rb").read()
    open(os.path.join(OUT, "refused_truncated.jpg"), "wb").write(good[: len(good) // 2])
    np.savez_compressed(os.path.join(OUT, "expected_rgb.npz"), **expected)
    print(f"{len(cases)} fixtures written; decoder library: libjpeg-turbo as built into Pillow "
          f"(jpeglib {features.version('jpg')})")


if __name__ == "__main__":
    main()
