"""Tile culling (rt_set_tile_culling): the sweeps test every tile's triangles against a wave's
whole bundle of rays first and only pass the survivors on to the per-ray tests.  The results
must not change: the parity cases of the other GPU test files are run again with culling on
(against the CPU oracle and the golden frames), and culled frames are compared bit for bit with
the frames of the plain all-pairs sweeps."""
import numpy as np
import pytest
import torch

from raytracer_amd import kernel, scenes
from conftest import TAN_HALF

import test_gpu_edge as edge
import test_gpu_fullsize as fullsize
import test_gpu_parity as parity
import test_gpu_parts as parts
import test_gpu_stress as stress

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def culling_on():
    kernel.set_tile_culling(True)
    assert kernel.tile_culling()
    yield
    kernel.set_tile_culling(None)


def test_setting_and_environment(monkeypatch):
    kernel.set_tile_culling(None)
    monkeypatch.delenv("RT_AMD_CULL", raising=False)
    assert not kernel.tile_culling()            # off unless asked for
    monkeypatch.setenv("RT_AMD_CULL", "1")
    assert kernel.tile_culling()
    kernel.set_tile_culling(False)              # the call overrides the environment
    assert not kernel.tile_culling()
    kernel.set_tile_culling(True)


@pytest.mark.parametrize("kind,tris,W,H,frame_no", [("soup", 100_000, 1920, 1080, 0), ("torus", 69_451, 1280, 720, 5),
                                                    ("soup", 3000, 333, 77, 11), ("sphere", 20_000, 640, 480, 3)])
def test_culled_frames_equal_all_pairs_frames(kind, tris, W, H, frame_no):
    """Every output of a whole shaded frame, bit for bit, with and without culling; the culled
    sweeps test far fewer pairs one by one."""
    mesh = {"soup": lambda: scenes.triangle_soup(tris, seed=77), "torus": lambda: scenes.torus(tris),
            "sphere": lambda: scenes.sphere(tris)}[kind]()
    fr = kernel.driver_frame(frame_no)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    want = ("hit", "tuv", "argb", "rgb8")
    kernel.profile_enable(True)
    try:
        kernel.profile_reset()
        culled = {k: v.clone() for k, v in sc.render_rows(fr, W, H, TAN_HALF, want=want).items()}
        tests_culled = kernel.profile_get().shadow_tests
        kernel.set_tile_culling(False)
        kernel.profile_reset()
        plain = sc.render_rows(fr, W, H, TAN_HALF, want=want)
        tests_plain = kernel.profile_get().shadow_tests
    finally:
        kernel.profile_enable(False)
    assert (plain["hit"] >= 0).float().mean() > 0.05
    for k in want:
        a, b = culled[k], plain[k]
        if a.dtype.is_floating_point:           # unhit pixels carry NaN-free fill values; compare bits
            a, b = a.view(torch.int64), b.view(torch.int64)
        assert torch.equal(a, b), k
    assert tests_culled < tests_plain
    if kind == "soup" and tris >= 100_000:
        assert tests_culled < tests_plain / 4
    sc.close()


# ---- the other files' cases again, culling on (each compares with the oracle itself)

@pytest.mark.parametrize("W,H", [(1, 1), (65, 17), (129, 33), (31, 250)])
def test_ragged_image_sizes(W, H):
    edge.test_ragged_image_sizes(W, H)


@pytest.mark.parametrize("ntris", [1, 127, 129, 257])
def test_triangle_counts_around_the_tile_size(ntris):
    edge.test_triangle_counts_around_the_tile_size(ntris)


def test_degenerate_overflow_textures():
    edge.test_degenerate_and_duplicate_triangles()
    edge.test_candidate_list_overflow_falls_back_to_full_fp64_sweep()
    edge.test_textured_materials()
    edge.test_empty_inputs_and_zero_rays()


@pytest.mark.parametrize("nl", [0, 1, 11])
def test_light_counts(nl):
    edge.test_light_counts(nl)


@pytest.mark.parametrize("dist", [300.0, 3.0e5])
def test_far_ray_origins(dist):
    edge.test_far_ray_origins(dist)


@pytest.mark.parametrize("through_eye", [3, 40])
def test_triangles_whose_plane_contains_the_eye(through_eye):
    edge.test_triangles_whose_plane_contains_the_eye(through_eye)


@pytest.mark.parametrize("shape", ["16,3", "8,2", "2,5", "1,1", "16,2,2", "32,3,1", "64,1", "64,3,4", "64,2,0", "32,5,0"])
@pytest.mark.parametrize("W,H", [(500, 20), (37, 19)])
def test_every_shape_of_the_pixel_grid_sweep(shape, W, H, monkeypatch):
    edge.test_every_shape_of_the_pixel_grid_sweep(shape, W, H, monkeypatch, None)


@pytest.mark.parametrize("shape", ["8,1", "8,4", "4,3", "2,2", "1,7"])
def test_every_shape_of_the_any_hit_sweep(shape, monkeypatch):
    edge.test_every_shape_of_the_any_hit_sweep(shape, monkeypatch)


@pytest.mark.parametrize("apex_shape", [None, "8,3", "2,5"])
def test_ranged_any_hit_sweeps_on_a_small_scene(apex_shape, monkeypatch):
    edge.test_ranged_any_hit_sweeps_on_a_small_scene(apex_shape, monkeypatch)


@pytest.mark.parametrize("W,H,n,block,tris", [(320, 200, 3, 16, 20000), (200, 77, 4, 8, 3000)])
def test_parts_equal_whole(W, H, n, block, tris):
    parts.test_parts_equal_whole(W, H, n, block, tris)


@pytest.mark.parametrize("name", sorted(parity.GOLDEN))
def test_golden_frames(name):
    parity.test_golden_frames(name)


def test_rays_edges_vertices_and_hard_geometry():
    parity.test_edge_and_vertex_aimed_rays()
    parity.test_band_rows_match_full_frame()
    stress.test_hard_geometry_has_no_mismatch()


def test_production_scale_against_the_oracle():
    """Culling at the sizes the benchmark runs: the shaded band of the 1 M-triangle 4K frame
    against the oracle (occluder cache, sorted rays, ranged sweeps with compaction in between),
    and the whole 1080p and 4K golden frames."""
    fullsize._shaded_rows_vs_oracle(scenes.triangle_soup(1_000_000, seed=12345), 0, 3840, 2160,
                                    band=(1000, 1064), check_rows=(1003, 1050), min_shadow_launches=8)
    fullsize.test_config2_whole_1080p_frame_against_the_oracles_golden()
    fullsize.test_config3_whole_4k_frame_against_the_oracles_golden()
