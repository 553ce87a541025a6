"""rt_render_sequence: the pipelined frame loop writes the same pictures as the frame-by-frame
calls, for a plain run and for a strided (frame-parallel) one."""
import os

import numpy as np
import pytest
import torch

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
