"""Writes the PNG fixtures of tests/test_png_textures.py and the RGB each must decode to.

The encoder below is written from the PNG specification (ISO/IEC 15948) for this purpose: every
colour type and bit depth, all five scanline filters (row y uses filter y % 5), optional Adam7
interlacing.  The expected pixels are derived from the source samples directly -- palette entries
resolved, grey replicated (1/2/4-bit grey scaled to full range), alpha dropped, 16-bit samples cut
to their high byte: what SDL_GetRGB reports for a surface SDL_image loaded (graphic_tools.c:161-197
reads texels that way) -- never from a decoder.  Where Pillow is importable the script also checks
that Pillow decodes every 8-bit-or-less fixture to the same RGB.

    python tests/golden/make_png_fixtures.py
"""
import os
import struct
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "png")
ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def chunk(kind: bytes, body: bytes) -> bytes:
    return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)


def pack_rows(samples: np.ndarray, depth: int) -> list:
    """samples: [h, w, channels] integers < 2**depth -> one bytes object per row, MSB first."""
    rows = []
    for r in samples:
        flat = r.reshape(-1)
        if depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in flat))
        elif depth == 8:
            rows.append(bytes(int(v) for v in flat))
        else:
            bits = "".join(format(int(v), f"0{depth}b") for v in flat)
            bits += "0" * (-len(bits) % 8)
            rows.append(bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
    return rows


def filter_rows(rows: list, bpp: int) -> bytes:
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, cur in enumerate(rows):
        f = y % 5
        out.append(f)
        for i, x in enumerate(cur):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                pred = 0
            elif f == 1:
                pred = a
            elif f == 2:
                pred = b
            elif f == 3:
                pred = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out.append((x - pred) & 0xFF)
        prev = cur
    return bytes(out)


def encode(samples: np.ndarray, ctype: int, depth: int, palette=None, interlace=False) -> bytes:
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    if interlace:
        raw = b""
        for x0, y0, dx, dy in ADAM7:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filter_rows(pack_rows(sub, depth), bpp)
    else:
        raw = filter_rows(pack_rows(samples, depth), bpp)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        png += chunk(b"PLTE", bytes(int(v) for v in np.asarray(palette).reshape(-1)))
    z = zlib.compress(raw, 9)
    half = len(z) // 2          # two IDAT chunks: the stream may be split anywhere
    return png + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"tEXt", b"Comment\0fixture") + chunk(b"IEND", b"")


def expected_rgb(samples, ctype, depth, palette):
    s = samples.astype(np.int64)
    if ctype == 3:
        return np.asarray(palette, np.uint8)[s[..., 0]]
    if depth == 16:
        s = s >> 8
    elif depth < 8 and ctype == 0:
        s = s * 255 // ((1 << depth) - 1)
    if ctype in (0, 4):
        return np.repeat(s[..., :1], 3, axis=2).astype(np.uint8)
    return s[..., :3].astype(np.uint8)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(2024)
    cases = [   # name, ctype, depth, channels, w, h, interlace
        ("rgb8", 2, 8, 3, 7, 5, False), ("rgba8", 6, 8, 4, 9, 6, False), ("grey8", 0, 8, 1, 11, 7, False),
        ("grey_alpha8", 4, 8, 2, 5, 9, False), ("pal8", 3, 8, 1, 13, 6, False), ("pal4", 3, 4, 1, 7, 7, False),
        ("pal2", 3, 2, 1, 9, 5, False), ("pal1", 3, 1, 1, 10, 6, False), ("grey4", 0, 4, 1, 5, 5, False),
        ("grey1", 0, 1, 1, 17, 3, False), ("rgb16", 2, 16, 3, 6, 5, False), ("rgba16", 6, 16, 4, 4, 6, False),
        ("grey16", 0, 16, 1, 7, 4, False), ("rgb8_adam7", 2, 8, 3, 13, 11, True), ("pal4_adam7", 3, 4, 1, 9, 10, True),
        ("rgba8_adam7_small", 6, 8, 4, 3, 2, True), ("rgb8_32x16", 2, 8, 3, 32, 16, False),
    ]
    expected = {}
    for name, ctype, depth, ch, w, h, inter in cases:
        samples = rng.integers(0, 1 << depth, size=(h, w, ch))
        palette = rng.integers(0, 256, size=(1 << depth, 3)) if ctype == 3 else None
        open(os.path.join(OUT, name + ".png"), "wb").write(encode(samples, ctype, depth, palette, inter))
        expected[name] = expected_rgb(samples, ctype, depth, palette)
    # files a decoder must refuse: truncated stream, bad signature
    good = open(os.path.join(OUT, "rgb8.png"), "rb").read()
    open(os.path.join(OUT, "bad_truncated.png"), "wb").write(good[:len(good) - 40])
    open(os.path.join(OUT, "bad_signature.png"), "wb").write(b"\x89PNX" + good[4:])
    np.savez_compressed(os.path.join(OUT, "expected_rgb.npz"), **expected)
    try:
        from PIL import Image
    except ImportError:
        print("Pillow not importable: fixtures written, independent decode check skipped")
        return
    checked = 0
    for name, ctype, depth, *_ in cases:
        if depth == 16:
            continue        # Pillow maps 16-bit samples to 8 bits by its own rule
        got = np.asarray(Image.open(os.path.join(OUT, name + ".png")).convert("RGB"))
        assert np.array_equal(got, expected[name]), name
        checked += 1
    print(f"{len(cases)} fixtures written, {checked} confirmed with Pillow")


if __name__ == "__main__":
    main()
