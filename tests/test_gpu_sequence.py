"""rt_render_sequence: the pipelined frame loop writes the same pictures as the frame-by-frame
calls, for a plain run and for a strided (frame-parallel) one, and the pictures it writes are
the oracle's (driver.c:28-203 restated on the CPU: frame set-up, transform, pixel loop, 8-bit
store, placement)."""
import os
import struct

import numpy as np
import pytest
import torch

from oracle import oracle as O
from raytracer_amd import kernel, scenes
from conftest import TAN_HALF

pytestmark = pytest.mark.gpu


def _frame_bmp(sc, frame_no, turn, W, H, path):
    fr = kernel.driver_frame(frame_no, turn)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF, want=("rgb8",))
    screen = sc.place_rows(out["rgb8"], W, H, 0, H)
    kernel.save_bmp(path, screen.cpu().numpy())


@pytest.mark.parametrize("first,count,step", [(0, 9, 1), (2, 3, 4)])
def test_sequence_writes_the_frames_of_the_single_frame_path(tmp_path, first, count, step):
    W, H, turn = 160, 120, 36
    mesh = scenes.with_materials(scenes.torus(1500))
    sc = kernel.Scene(mesh)
    st = sc.render_sequence(first, count, W, H, TAN_HALF, frames_per_turn=turn, frame_step=step,
                            bmp_pattern=str(tmp_path / "pic%04d.bmp"))
    assert st.frames == count and st.seconds_total > 0 and st.seconds_writing > 0
    frames = [first + k * step for k in range(count)]
    assert sorted(os.listdir(tmp_path)) == [f"pic{f:04d}.bmp" for f in frames]
    for f in frames:
        ref = str(tmp_path / f"ref{f}.bmp")
        _frame_bmp(sc, f, turn, W, H, ref)
        a, b = open(tmp_path / f"pic{f:04d}.bmp", "rb").read(), open(ref, "rb").read()
        assert len(a) == 122 + W * H * 4 and a == b, f
    # pictures differ from frame to frame (the orbit moves)
    assert open(tmp_path / f"pic{frames[0]:04d}.bmp", "rb").read() != open(tmp_path / f"pic{frames[1]:04d}.bmp", "rb").read()
    sc.close()


def _decode_bmp(path, W, H):
    """The 32-bit BMP of graphic_tools.c:19-23,115-121 -> H x W x 3 RGB, top row first."""
    raw = open(path, "rb").read()
    off = struct.unpack_from("<I", raw, 10)[0]
    w, h = struct.unpack_from("<ii", raw, 18)
    assert (w, h) == (W, H) and struct.unpack_from("<H", raw, 28)[0] == 32
    px = np.frombuffer(raw, dtype=np.uint8, offset=off).reshape(H, W, 4)[::-1]      # bottom-up rows, B G R A
    assert np.all(px[..., 3] == 255)
    return px[..., 2::-1]


def test_sequence_pictures_are_the_oracles(tmp_path):
    """Frames of the written sequence against the CPU oracle's whole frame loop: the oracle's own
    frame set-up (pinned to the reference's M3d code), its transform, pixel loop and placement."""
    W, H, turn = 128, 96, 24
    mesh = scenes.with_materials(scenes.torus(1200))
    sc = kernel.Scene(mesh)
    frames = [1, 6, 11, 16, 21]
    st = sc.render_sequence(frames[0], len(frames), W, H, TAN_HALF, frames_per_turn=turn, frame_step=5,
                            bmp_pattern=str(tmp_path / "pic%04d.bmp"))
    assert st.frames == len(frames)
    lit = 0
    for f in frames:
        got = _decode_bmp(tmp_path / f"pic{f:04d}.bmp", W, H)
        ofr = O.frame_setup(f, turn)                                  # driver.c:34-107
        X, Y, Z = O.transform(mesh, O.arr4(ofr.obmat))                # driver.c:110
        ref = O.render_rows(O.Scene(mesh, X, Y, Z), ofr, W, H, TAN_HALF)      # driver.c:126-185
        want = O.place_pixels(ref["rgb8"])                            # driver.c:186-190
        assert (ref["hit"] >= 0).sum() > 0.05 * W * H
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 1, f"frame {f}: a channel differs by {d.max()}"     # 1/255 per channel
        assert (d > 0).mean() < 1e-3
        assert np.all(got[0] == 0)                                    # row 0 keeps the clear colour
        lit += len(np.unique(want.reshape(-1, 3), axis=0))
    assert lit > 50 * len(frames)                                     # shaded pictures, not flat ones
    sc.close()


def test_sequence_without_files_and_bad_arguments(tmp_path):
    import ctypes as C
    from raytracer_amd import _lib
    sc = kernel.Scene(scenes.torus(300))
    st = sc.render_sequence(0, 3, 64, 48, TAN_HALF)
    assert st.frames == 3 and st.seconds_writing == 0
    lib = _lib.load()
    assert lib.rt_render_sequence(sc.h, 0, 2, 0, 60, 64, 48, TAN_HALF, None, None) == -1      # step 0
    assert lib.rt_render_sequence(sc.h, 0, 2, 1, 60, 0, 48, TAN_HALF, None, None) == -1       # W 0
    missing = str(tmp_path / "no_such_dir" / "pic%04d.bmp").encode()
    assert lib.rt_render_sequence(sc.h, 0, 2, 1, 60, 64, 48, TAN_HALF, missing, None) == -4   # RT_E_IO
    assert b"cannot open" in lib.rt_last_error()
    sc.close()
