"""rt_render_frame_multi: one frame split over several devices inside one process.  A one-GPU
box can run every step but an exchange between two different GPUs: one device (no exchange),
the same device named several times (threads, parts, placement; the rows move by peer copies,
because RCCL refuses a communicator with duplicate devices), and one device with the exchange
forced through RCCL (a grouped ncclSend / ncclRecv of the part to itself).  The picture must
equal the single-scene render, placed, bit for bit, and the oracle's within 1/255."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from raytracer_amd import _lib, kernel, scenes
from conftest import TAN_HALF

pytestmark = pytest.mark.gpu


def _single(mesh, fr, W, H):
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF, want=("rgb8",))
    screen = sc.place_rows(out["rgb8"], W, H, 0, H)
    sc.close()
    return screen


@pytest.mark.parametrize("devices,exchange", [([0], None), ([0, 0], None), ([0, 0, 0, 0, 0], None), ([0], "rccl")])
def test_multi_frame_equals_single_scene_frame(devices, exchange, monkeypatch):
    if exchange:
        monkeypatch.setenv("RT_AMD_MULTI_EXCHANGE", exchange)
    mesh = scenes.with_materials(scenes.torus(3000))
    W, H = 200, 150                   # 150 rows: the last round of 16-row blocks is a partial one
    ms = kernel.MultiScene(mesh, devices)
    assert ms.exchange == ("rccl" if exchange == "rccl" else "copy")
    for frame_no in (0, 7):
        fr = kernel.driver_frame(frame_no)
        ms.transform(kernel.frame_matrix(fr, "obmat"))
        screen = ms.render_frame(fr, W, H, TAN_HALF)
        assert torch.equal(screen, _single(mesh, fr, W, H)), (devices, frame_no)
    # ... and the oracle's picture for the last frame
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    ref = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF)
    d = np.abs(screen.cpu().numpy().astype(int) - O.place_pixels(ref["rgb8"]).astype(int))
    assert d.max() <= 1
    ms.close()


@pytest.mark.parametrize("exchange", ["rccl", "copy"])
def test_two_different_gpus(exchange, monkeypatch):
    """The exchange between two different GPUs, over RCCL (grouped ncclSend / ncclRecv on two
    communicators and streams) and by peer copies: runs wherever a node with two GPUs exists."""
    if _lib.load().rt_device_count() < 2:
        pytest.skip("needs two GPUs")
    monkeypatch.setenv("RT_AMD_MULTI_EXCHANGE", exchange)
    mesh = scenes.with_materials(scenes.torus(3000))
    W, H = 200, 150
    ms = kernel.MultiScene(mesh, [0, 1])
    assert ms.exchange == exchange
    for frame_no in (0, 7):
        fr = kernel.driver_frame(frame_no)
        ms.transform(kernel.frame_matrix(fr, "obmat"))
        assert torch.equal(ms.render_frame(fr, W, H, TAN_HALF), _single(mesh, fr, W, H)), frame_no
    ms.close()


def test_many_frames_on_the_same_worker_threads():
    """The devices' host threads live as long as the set: many calls in a row, and a second set
    beside the first."""
    mesh = scenes.torus(1200)
    W, H = 96, 80
    a, b = kernel.MultiScene(mesh, [0, 0, 0]), kernel.MultiScene(mesh, [0, 0])
    for frame_no in range(12):
        fr = kernel.driver_frame(frame_no)
        want = _single(mesh, fr, W, H)
        for ms in (a, b):
            ms.transform(kernel.frame_matrix(fr, "obmat"))
            assert torch.equal(ms.render_frame(fr, W, H, TAN_HALF), want), frame_no
    a.close()
    b.close()


def test_multi_more_parts_than_rows_and_errors():
    lib = _lib.load()
    mesh = scenes.torus(400)
    fr = kernel.driver_frame(2)
    ms = kernel.MultiScene(mesh, [0, 0, 0])
    ms.transform(kernel.frame_matrix(fr, "obmat"))
    W, H = 64, 2                      # fewer rows than parts: some parts are empty
    assert torch.equal(ms.render_frame(fr, W, H, TAN_HALF), _single(mesh, fr, W, H))
    screen = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    assert lib.rt_render_frame_multi(ms.h, C.byref(fr), 0, H, TAN_HALF, screen.data_ptr()) == -1
    assert lib.rt_render_frame_multi(ms.h, C.byref(fr), W, H, TAN_HALF, None) == -1
    ms.close()
    h = C.c_void_p()
    bad = np.array([0, 99], np.int32)
    assert lib.rt_multi_create(C.byref(h), 2, bad.ctypes.data_as(_lib.c_int_p)) == -1
    assert lib.rt_multi_create(C.byref(h), 0, None) == -1
    # a scene that was never transformed: the device's own error comes back with its part number
    ms = kernel.MultiScene(mesh, [0, 0])
    assert lib.rt_render_frame_multi(ms.h, C.byref(fr), W, 48, TAN_HALF, screen.data_ptr()) == -3
    assert b"rt_scene_transform" in lib.rt_last_error() and b"part" in lib.rt_last_error()
    ms.close()
