"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Hit indices and t/u/v must be bit-exact (the confirmation pass evaluates the
reference's fp64 formulas); colours must match to the 8-bit store within 1/255."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import kernel, scenes
from conftest import TAN_HALF

pytestmark = pytest.mark.gpu


def _oracle_eye(mesh, frame):
    X, Y, Z = O.transform(mesh, O.arr4(frame.obmat))
    return O.Scene(mesh, X, Y, Z), (X, Y, Z)


def _meshes():
    return [
        scenes.triangle_soup(300, seed=1),
        scenes.triangle_soup(5000, seed=2),
        scenes.torus(1200),
        scenes.sphere(900),
        scenes.with_materials(scenes.torus(700), num_mats=3),
    ]


@pytest.mark.parametrize("mi", range(5))
def test_transform_bit_exact(mi):
    mesh = _meshes()[mi]
    fr = kernel.driver_frame(2)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    gx, gy, gz = sc.eye_vertices()
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    assert np.array_equal(gx, X) and np.array_equal(gy, Y) and np.array_equal(gz, Z)
    sc.close()


@pytest.mark.parametrize("mi,W,H", [(0, 64, 48), (1, 96, 64), (2, 160, 120), (3, 100, 75), (4, 64, 64)])
def test_primary_hits_bit_exact(mi, W, H):
    mesh = _meshes()[mi]
    fr = kernel.driver_frame(1)
    osc, _ = _oracle_eye(mesh, O.frame_from(fr))
    ref = O.render_rows(osc, O.frame_from(fr), W, H, TAN_HALF, shade=False)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    hit, tuv = sc.trace_primary(W, H, TAN_HALF)
    hit = hit.cpu().numpy()
    tuv = tuv.cpu().numpy()
    assert (ref["hit"] >= 0).sum() > 0.02 * W * H, "scene misses the camera; test is vacuous"
    assert np.array_equal(hit, ref["hit"])
    m = ref["hit"] >= 0
    assert np.array_equal(tuv[m], ref["tuv"][m])
    sc.close()


@pytest.mark.parametrize("mi,W,H", [(0, 64, 48), (2, 128, 96), (4, 80, 60)])
def test_render_rows_rgb(mi, W, H):
    mesh = _meshes()[mi]
    fr = kernel.driver_frame(5)
    osc, _ = _oracle_eye(mesh, O.frame_from(fr))
    ref = O.render_rows(osc, O.frame_from(fr), W, H, TAN_HALF, shade=True)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF)
    hit = out["hit"].cpu().numpy()
    rgb8 = out["rgb8"].cpu().numpy()
    argb = out["argb"].cpu().numpy()
    assert np.array_equal(hit, ref["hit"])
    # tolerance stated by the task: 1/255 per channel after the 8-bit store
    diff = np.abs(rgb8.astype(int) - ref["rgb8"].astype(int))
    assert diff.max() <= 1, f"max channel diff {diff.max()}"
    assert np.abs(argb - ref["argb"]).max() < 1e-9
    assert ref["shadow_rays"] > 0
    sc.close()


def test_band_rows_match_full_frame():
    mesh = scenes.torus(800)
    fr = kernel.driver_frame(0)
    W, H = 96, 70
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    full = sc.render_rows(fr, W, H, TAN_HALF)
    parts = []
    for r in range(3):
        y0, y1 = kernel.band(H, 3, r)
        parts.append(sc.render_rows(fr, W, H, TAN_HALF, y0, y1))
    import torch
    for k in ("hit", "rgb8"):
        assert torch.equal(torch.cat([p[k] for p in parts], 0), full[k])
    sc.close()


def test_trace_rays_and_occluded_match_oracle():
    mesh = scenes.triangle_soup(2000, seed=9)
    fr = kernel.driver_frame(0)
    osc, (X, Y, Z) = _oracle_eye(mesh, O.frame_from(fr))
    rng = np.random.default_rng(3)
    n = 3000
    c = np.array([X[1:].mean(), Y[1:].mean(), Z[1:].mean()])
    S = c + rng.uniform(-12, 12, size=(n, 3))
    E = c + rng.uniform(-6, 6, size=(n, 3))
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    hit, tuv, point = sc.trace_rays(S, E)
    occ = sc.occluded(S, E)
    hit, tuv, point, occ = hit.cpu().numpy(), tuv.cpu().numpy(), point.cpu().numpy(), occ.cpu().numpy()
    nh = 0
    for k in range(n):
        idx, t, uv, p = O.intersect(osc, S[k], E[k])
        assert hit[k] == idx
        assert occ[k] == (1 if idx != -1 else 0)
        if idx != -1:
            nh += 1
            assert tuv[k, 0] == t and tuv[k, 1] == uv[0] and tuv[k, 2] == uv[1]
            assert np.array_equal(point[k], p)
    assert nh > n // 20
    sc.close()


def test_edge_and_vertex_aimed_rays():
    """Rays aimed exactly at shared edges and vertices of a closed mesh: the fp32 filter must
    hand every such pair to the fp64 confirmation, whose tie rules then decide."""
    mesh = scenes.torus(600)
    fr = kernel.driver_frame(4)
    osc, (X, Y, Z) = _oracle_eye(mesh, O.frame_from(fr))
    rng = np.random.default_rng(11)
    tri = rng.integers(0, mesh.num_tris, size=1500)
    A = np.stack([X[mesh.A[tri]], Y[mesh.A[tri]], Z[mesh.A[tri]]], 1)
    B = np.stack([X[mesh.B[tri]], Y[mesh.B[tri]], Z[mesh.B[tri]]], 1)
    Cc = np.stack([X[mesh.C[tri]], Y[mesh.C[tri]], Z[mesh.C[tri]]], 1)
    w = rng.uniform(0, 1, size=(len(tri), 1))
    targets = np.concatenate([A, w * A + (1 - w) * B, w * B + (1 - w) * Cc, w * A + (1 - w) * Cc])
    S = np.zeros_like(targets)
    E = targets * 1.0
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    hit, tuv, _ = sc.trace_rays(S, E)
    hit = hit.cpu().numpy()
    bad = 0
    for k in range(len(S)):
        idx, t, uv, p = O.intersect(osc, S[k], E[k])
        bad += int(hit[k] != idx)
    assert bad == 0, f"{bad} of {len(S)} edge-aimed rays differ"
    sc.close()


def test_legacy_kernel_h_api():
    mesh = scenes.with_materials(scenes.sphere(500))
    fr = kernel.driver_frame(7)
    obmat = kernel.frame_matrix(fr, "obmat")
    x, y, z = mesh.x.copy(), mesh.y.copy(), mesh.z.copy()
    kernel.mat_mult_device(x, y, z, obmat, x, y, z, mesh.num_v + 1)      # driver.c:110
    X, Y, Z = O.transform(mesh, obmat)
    assert np.array_equal(x, X) and np.array_equal(y, Y) and np.array_equal(z, Z)
    osc = O.Scene(mesh, X, Y, Z)
    tris = kernel.TriangleArray(mesh)
    W, H = 24, 18
    nh = 0
    for xp in range(W):
        for yp in range(H):
            S = np.zeros(3)
            E = np.array([xp - W // 2, yp - H // 2, (W // 2) / TAN_HALF])
            uv, point = np.full(2, -7.0), np.full(3, -7.0)
            s = kernel.intersect_all_triangles_device(S, E, uv, point, tris, x, y, z, mesh.num_tris)
            idx, t, ouv, op = O.intersect(osc, S, E)
            assert s == idx
            if idx == -1:
                assert np.all(uv == -7.0) and np.all(point == -7.0)      # untouched on a miss
            else:
                nh += 1
                assert np.array_equal(uv, ouv) and np.array_equal(point, op)
    assert nh > 10
    # in-place inverse transform must invalidate the cached device mesh (driver.c:202)
    obinv = kernel.frame_matrix(fr, "obinv")
    kernel.mat_mult_device(x, y, z, obinv, x, y, z, mesh.num_v + 1)
    osc2 = O.Scene(mesh, x.copy(), y.copy(), z.copy())
    S, E = np.array([0.0, 0.0, 30.0]), np.array([0.0, 0.0, -30.0])
    uv, point = np.zeros(2), np.zeros(3)
    s = kernel.intersect_all_triangles_device(S, E, uv, point, tris, x, y, z, mesh.num_tris)
    idx, t, ouv, op = O.intersect(osc2, S, E)
    assert s == idx and idx != -1 and np.array_equal(uv, ouv)


GOLDEN = {
    "frame_soup400_64x48_f3": (lambda: scenes.triangle_soup(400, seed=21), 3, 64, 48),
    "frame_torus600_80x60_f11": (lambda: scenes.with_materials(scenes.torus(600), num_mats=3, seed=5), 11, 80, 60),
    "frame_sphere300_48x64_f40": (lambda: scenes.sphere(300), 40, 48, 64),
}


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_golden_frames(name):
    """Committed fixtures (tests/golden/make_golden.py): the GPU box has no reference checkout
    and needs no oracle run for these."""
    import os
    make, frame_no, W, H = GOLDEN[name]
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    mesh = make()
    fr = kernel.driver_frame(frame_no)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF)
    hit = out["hit"].cpu().numpy()
    assert np.array_equal(hit, g["hit"])
    m = g["hit"] >= 0
    assert np.array_equal(out["tuv"].cpu().numpy()[m], g["tuv"][m])
    assert np.abs(out["rgb8"].cpu().numpy().astype(int) - g["rgb8"].astype(int)).max() <= 1
    assert np.abs(out["argb"].cpu().numpy() - g["argb"]).max() < 1e-9
    screen = sc.place_rows(out["rgb8"], W, H, 0, H).cpu().numpy()
    assert np.array_equal(screen, O.place_pixels(out["rgb8"].cpu().numpy()))
    sc.close()


def test_obj_file_to_frame(tmp_path):
    """Loader -> scene -> frame: rt_obj_load + rt_scene_set_obj against the oracle fed by the
    Python restatement of the parser."""
    from oracle import obj_oracle
    from raytracer_amd import _lib
    src = scenes.with_materials(scenes.torus(300, normalize=False), num_mats=2, unassigned_every=4)
    (tmp_path / "m.mtl").write_text("".join(
        f"newmtl m{k}\nKa {' '.join(repr(float(v)) for v in src.Ka[k])}\n"
        f"Kd {' '.join(repr(float(v)) for v in src.Kd[k])}\nKs {' '.join(repr(float(v)) for v in src.Ks[k])}\n"
        for k in range(2)))
    lines = ["mtllib m.mtl"] + scenes.to_obj_text(src, fmt=4, quads=False).splitlines()
    body, cur = [], None
    fi = 0
    for ln in lines:
        if ln.startswith("f "):
            m = int(src.mtl[fi]); fi += 1
            if m >= 0 and m != cur:
                body.append(f"usemtl m{m}"); cur = m
        body.append(ln)
    (tmp_path / "m.obj").write_text("\n".join(body) + "\n")
    ref = obj_oracle.load(str(tmp_path / "m.obj"))
    mesh = scenes.Mesh(**{k: np.ascontiguousarray(ref[k]) for k in
                          ("x", "y", "z", "xn", "yn", "zn", "A", "B", "C", "An", "Bn", "Cn")},
                       mtl=np.ascontiguousarray(ref["mtl"]), Ka=ref["Ka"], Kd=ref["Kd"], Ks=ref["Ks"])
    fr = kernel.driver_frame(9)
    W, H = 72, 54
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    want = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF)
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rt_obj_load(str(tmp_path / "m.obj").encode(), C.byref(h)) == 0
    sc = kernel.Scene()
    _lib.check(lib.rt_scene_set_obj(sc.h, h), "rt_scene_set_obj")
    v = _lib.RtObjView()
    lib.rt_obj_get(h, C.byref(v))
    assert v.pinned == 1                       # page-locked SoA buffers when a GPU is present
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF)
    assert np.array_equal(out["hit"].cpu().numpy(), want["hit"])
    assert np.abs(out["rgb8"].cpu().numpy().astype(int) - want["rgb8"].astype(int)).max() <= 1
    assert (want["hit"] >= 0).sum() > 100
    lib.rt_obj_free(h)
    sc.close()
