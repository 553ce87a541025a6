"""The cell-lattice any-hit sweep of the shadow rays (rt_sweep_cells.hip).

The plane of directions through a light is cut into a grid of cells, every (cell, triangle) pair
gets a separable first test, and behind it every ray of a passing cell is tested on its own with
the arithmetic of the ray-by-ray sweep.  The results must not depend on which of the two sweeps
ran: frames rendered with the lattice forced on are compared bit for bit with frames rendered
with it forced off, for every kernel shape and for coarse and fine grids, and the other files'
oracle and golden cases are run again with the lattice forced on (each compares with the CPU
oracle itself)."""
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
def lattice_on(monkeypatch):
    monkeypatch.setenv("RT_AMD_SHADOW_LATTICE", "1")
    yield


MESHES = {"soup": lambda n: scenes.triangle_soup(n, seed=77), "torus": scenes.torus, "sphere": scenes.sphere}


def render(sc, fr, W, H, want=("hit", "tuv", "argb", "rgb8")):
    return {k: v.clone() for k, v in sc.render_rows(fr, W, H, TAN_HALF, want=want).items()}


def same(a, b):
    for k in a:
        x, y = a[k], b[k]
        if x.dtype.is_floating_point:
            x, y = x.view(torch.int64), y.view(torch.int64)
        assert torch.equal(x, y), k


@pytest.mark.parametrize("kind,tris,W,H,frame_no", [("soup", 100_000, 1920, 1080, 0), ("torus", 69_451, 1280, 720, 5),
                                                    ("soup", 3000, 333, 77, 11), ("sphere", 20_000, 640, 480, 3),
                                                    ("torus", 6_320, 640, 480, 40)])
def test_lattice_frames_equal_ray_by_ray_frames(kind, tris, W, H, frame_no, monkeypatch):
    """Every output of a whole shaded frame, bit for bit, with the lattice and without; the lattice
    really ran (its launches are counted) and tested far fewer pairs one by one."""
    mesh = MESHES[kind](tris)
    fr = kernel.driver_frame(frame_no)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    kernel.profile_enable(True)
    try:
        kernel.profile_reset()
        lattice = render(sc, fr, W, H)
        p = kernel.profile_get()
        cell_launches, ray_pairs, cell_rays = p.cell_launches, p.cell_ray_pairs, p.cell_rays
        monkeypatch.setenv("RT_AMD_SHADOW_LATTICE", "0")
        kernel.profile_reset()
        plain = render(sc, fr, W, H)
        q = kernel.profile_get()
    finally:
        kernel.profile_enable(False)
    assert (plain["hit"] >= 0).float().mean() > 0.05
    assert cell_launches >= 1 and cell_rays > 0 and q.cell_launches == 0
    same(lattice, plain)
    assert ray_pairs < cell_rays * mesh.num_tris / 4
    sc.close()


@pytest.mark.parametrize("shape", ["2048", "2048,7", "512,3", "128,1", "128,40", "32,2", "8,1", "8,5"])
@pytest.mark.parametrize("cells_per_ray", ["0.02", "0.5", "16"])
@pytest.mark.parametrize("fine,sort", [("0.001", None), ("1", "1"), ("4096", None), ("1", "0")])
def test_every_shape_of_the_cell_sweep(shape, cells_per_ray, fine, sort, monkeypatch):
    """Every kernel shape (cells per lane, triangle chunks) on coarse and fine grids, with the
    rays looked up by the lattice's own cells (fine = 0.001 per ray), by 4 or 16 fine cells per
    cell, or by as many as the launch allows (256 per cell), and with the triangles in the lattice's fixed scattered
    order (the default there), in the order of their solid angle, or as the mesh lists them: equal to the ray-by-ray sweep, which equals the oracle
    in the other files."""
    mesh = scenes.with_materials(scenes.triangle_soup(2500, seed=5))
    fr = kernel.driver_frame(7)
    W, H = 210, 90
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    monkeypatch.setenv("RT_AMD_SHADOW_LATTICE", "0")
    plain = render(sc, fr, W, H)
    monkeypatch.setenv("RT_AMD_SHADOW_LATTICE", "1")
    monkeypatch.setenv("RT_AMD_CELLS_FORCE", shape)
    monkeypatch.setenv("RT_AMD_CELLS_PER_RAY", cells_per_ray)
    monkeypatch.setenv("RT_AMD_CELLS_FINE", fine)
    if sort:
        monkeypatch.setenv("RT_AMD_SORT_TRIS", sort)
    kernel.profile_enable(True)
    try:
        kernel.profile_reset()
        lattice = render(sc, fr, W, H)
        p = kernel.profile_get()
    finally:
        kernel.profile_enable(False)
    assert p.cell_launches >= 1 and p.cells_per_lane == int(shape.split(",")[0])
    assert (plain["hit"] >= 0).float().mean() > 0.3
    same(lattice, plain)
    sc.close()


@pytest.mark.parametrize("parts", ["1", "2", "4", "8"])
@pytest.mark.parametrize("shape", ["2048,20", "128,20", "8,20"])
@pytest.mark.parametrize("fine", [None, "4"])
def test_tiles_cut_in_parts(parts, shape, fine, monkeypatch):
    """A tile's triangles dealt to 1, 2, 4 or 8 workgroups (every wave sweeps 32, 16, 8 or 4 of a
    tile's triangles), with one tile per chunk as in the launches that choose the cut themselves,
    and with the fine cells the sweep picks itself or four per ray: equal to the ray-by-ray sweep."""
    mesh = scenes.with_materials(scenes.torus(2500))      # 20 tiles
    fr = kernel.driver_frame(11)
    W, H = 230, 110
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    monkeypatch.setenv("RT_AMD_SHADOW_LATTICE", "0")
    plain = render(sc, fr, W, H)
    monkeypatch.setenv("RT_AMD_SHADOW_LATTICE", "1")
    monkeypatch.setenv("RT_AMD_CELLS_FORCE", shape)
    monkeypatch.setenv("RT_AMD_CELLS_PARTS", parts)
    if fine:
        monkeypatch.setenv("RT_AMD_CELLS_FINE", fine)
    kernel.profile_enable(True)
    try:
        kernel.profile_reset()
        lattice = render(sc, fr, W, H)
        p = kernel.profile_get()
    finally:
        kernel.profile_enable(False)
    assert p.cell_launches >= 1 and p.cell_ray_pairs > 0
    assert (plain["hit"] >= 0).float().mean() > 0.2
    same(lattice, plain)
    sc.close()


def test_lattice_against_the_oracle_directly():
    """A shaded frame with the lattice on against the CPU oracle (not only against the other sweep)."""
    from oracle import oracle as O
    mesh = scenes.with_materials(scenes.torus(1500))
    fr = kernel.driver_frame(3)
    W, H = 160, 120
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    kernel.profile_enable(True)
    try:
        kernel.profile_reset()
        out = sc.render_rows(fr, W, H, TAN_HALF)
        assert kernel.profile_get().cell_launches >= 1
    finally:
        kernel.profile_enable(False)
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    ref = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF)
    assert np.array_equal(out["hit"].cpu().numpy(), ref["hit"])
    assert np.abs(out["argb"].cpu().numpy() - ref["argb"]).max() < 1e-9
    sc.close()


# ---- the other files' cases again, lattice forced on (each compares with the oracle itself)

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


@pytest.mark.parametrize("nl", [0, 1, 5, 11])
def test_light_counts(nl):
    edge.test_light_counts(nl)


@pytest.mark.parametrize("through_eye", [3, 40])
def test_triangles_whose_plane_contains_the_eye(through_eye):
    edge.test_triangles_whose_plane_contains_the_eye(through_eye)


def test_triangles_whose_plane_contains_a_light():
    edge.test_triangles_whose_plane_contains_a_light()


@pytest.mark.parametrize("W,H,tan_half", [(8192, 16, TAN_HALF), (3, 4096, TAN_HALF), (96, 64, 10.0), (96, 64, 0.01)])
def test_extreme_image_shapes_and_fields_of_view(W, H, tan_half):
    edge.test_extreme_image_shapes_and_fields_of_view(W, H, tan_half)


@pytest.mark.parametrize("W,H,n,block,tris", [(320, 200, 3, 16, 20000), (200, 77, 4, 8, 3000)])
def test_parts_equal_whole(W, H, n, block, tris):
    parts.test_parts_equal_whole(W, H, n, block, tris)


@pytest.mark.parametrize("name", sorted(parity.GOLDEN))
def test_golden_frames(name):
    parity.test_golden_frames(name)


def test_hard_geometry():
    parity.test_band_rows_match_full_frame()
    stress.test_hard_geometry_has_no_mismatch()
