"""Interleaved row parts on the GPU: rendering an image part by part gives bit for bit what the
whole-image call gives, for every output, and placing the parts builds the same screen."""
import numpy as np
import pytest
import torch

from raytracer_amd import kernel, scenes
from conftest import TAN_HALF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,parts,block,tris", [(320, 200, 3, 16, 20000), (200, 77, 4, 8, 3000),
                                                  (640, 64, 2, 16, 100000), (96, 21, 8, 16, 2000),
                                                  (96, 5, 8, 16, 2000)])
def test_parts_equal_whole(W, H, parts, block, tris):
    mesh = scenes.with_materials(scenes.torus(tris))
    fr = kernel.driver_frame(7)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    whole = {k: v.cpu().numpy() for k, v in sc.render_rows(fr, W, H, TAN_HALF).items()}
    assert (whole["hit"] >= 0).any()
    screen_whole = sc.place_rows(torch.as_tensor(whole["rgb8"], device="cuda"), W, H, 0, H).cpu().numpy()
    screen = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    for p in range(parts):
        rs = kernel.rows_part(H, parts, p, block)
        rows = kernel.rows_image_rows(rs)
        got = sc.render_rows(fr, W, H, TAN_HALF, rows_set=rs)
        for k in ("hit", "tuv", "argb", "rgb8"):
            g = got[k].cpu().numpy()
            hitmask = whole["hit"][rows] >= 0
            if k == "tuv":      # only written on a hit
                assert np.array_equal(g[hitmask], whole[k][rows][hitmask]), (k, p)
            else:
                assert np.array_equal(g, whole[k][rows]), (k, p)
        sc.place_rows(got["rgb8"], W, H, screen=screen, rows_set=rs)
    assert np.array_equal(screen.cpu().numpy(), screen_whole)
    sc.close()


def test_part_of_a_sub_range_and_primary_only():
    W, H = 160, 120
    mesh = scenes.torus(6000)
    fr = kernel.driver_frame(0)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    hit_all, _ = sc.trace_primary(W, H, TAN_HALF)
    rs = kernel.rows_part(H, 2, 1, 16, y0=20, y1=101)
    rows = kernel.rows_image_rows(rs)
    hit = torch.empty((len(rows), W), dtype=torch.int32, device="cuda")
    kernel.check(sc.lib.rt_trace_primary_part(sc.h, W, H, TAN_HALF, rs, hit.data_ptr(), None),
                 "rt_trace_primary_part")
    assert np.array_equal(hit.cpu().numpy(), hit_all.cpu().numpy()[rows])
    sc.close()


@pytest.mark.parametrize("kind,tris,parts", [("soup", 1_000_000, 8), ("soup", 1_000_000, 3), ("torus", 100_000, 8),
                                             ("sphere", 69_451, 4)])
def test_parts_of_the_4k_frames_equal_the_whole(kind, tris, parts):
    """BASELINE configurations 4, 2-class and 3 at full size, as several GPUs would render them:
    every interleaved part (its own launch shapes: up to 1024 triangle chunks in the primary
    sweep, 8 rays per lane and ranges that grow by 3x in the any-hit sweeps) gives bit for bit the
    rows of the whole-frame call (for the soup: 256 pixels per lane, 16-ray any-hit launches,
    ranges that grow by 2x)."""
    W, H = 3840, 2160
    mesh = {"soup": lambda: scenes.triangle_soup(tris, seed=12345), "torus": lambda: scenes.torus(tris),
            "sphere": lambda: scenes.sphere(tris)}[kind]()
    fr = kernel.driver_frame(0)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    want = ("hit", "rgb8", "argb")
    whole = {k: v.cpu().numpy() for k, v in sc.render_rows(fr, W, H, TAN_HALF, want=want).items()}
    assert (whole["hit"] >= 0).mean() > 0.1
    covered = np.zeros(H, dtype=bool)
    for p in range(parts):
        rs = kernel.rows_part(H, parts, p)
        rows = kernel.rows_image_rows(rs)
        covered[rows] = True
        got = sc.render_rows(fr, W, H, TAN_HALF, rows_set=rs, want=want)
        for k in want:
            assert np.array_equal(got[k].cpu().numpy(), whole[k][rows]), (k, p)
    assert covered.all()
    sc.close()
