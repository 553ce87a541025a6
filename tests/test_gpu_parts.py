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
