"""Edge cases of the HIP path through the C ABI: ragged image sizes, tiny and empty inputs,
degenerate geometry, candidate-list overflow, textures, light-count extremes, error codes."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as O
from raytracer_amd import _lib, kernel, scenes
from conftest import TAN_HALF

pytestmark = pytest.mark.gpu


def _ref(mesh, fr, W, H, **kw):
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    return O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF, **kw), (X, Y, Z)


def _gpu(mesh, fr, W, H, **kw):
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF, **kw)
    return sc, {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("W,H", [(1, 1), (7, 5), (65, 17), (70, 37), (129, 33), (31, 250)])
def test_ragged_image_sizes(W, H):
    """Sizes that are not multiples of the 64x16 workgroup tile, down to a single pixel."""
    mesh = scenes.torus(300)
    fr = kernel.driver_frame(2)
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert np.array_equal(out["hit"], ref["hit"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    sc.close()


@pytest.mark.parametrize("W,H,tan_half", [(8192, 16, TAN_HALF), (3, 4096, TAN_HALF), (96, 64, 10.0), (96, 64, 0.01),
                                          (640, 40, 1e-4)])
def test_extreme_image_shapes_and_fields_of_view(W, H, tan_half):
    """The bounding boxes of the first tests live on the image plane z = (W/2)/tan_half: very wide
    and very tall images, a field of view of 169 degrees (everything projects near the centre)
    and of a degree or less (the projection plane is thousands of pixels away and most of the mesh
    falls outside the image)."""
    for mesh in (scenes.torus(600), scenes.triangle_soup(500, seed=3)):
        fr = kernel.driver_frame(5)
        X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
        ref = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, tan_half)
        sc = kernel.Scene(mesh)
        sc.transform(kernel.frame_matrix(fr, "obmat"))
        for cull in (False, True):
            kernel.set_tile_culling(cull)
            try:
                out = {k: v.cpu().numpy() for k, v in sc.render_rows(fr, W, H, tan_half).items()}
            finally:
                kernel.set_tile_culling(None)
            assert np.array_equal(out["hit"], ref["hit"]), (mesh.name, cull)
            assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
        sc.close()


@pytest.mark.parametrize("ntris", [1, 2, 127, 128, 129, 257])
def test_triangle_counts_around_the_tile_size(ntris):
    mesh = scenes.triangle_soup(ntris, seed=ntris)
    fr = kernel.driver_frame(0)
    W, H = 80, 60
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert np.array_equal(out["hit"], ref["hit"])
    m = ref["hit"] >= 0
    assert np.array_equal(out["tuv"][m], ref["tuv"][m])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    sc.close()


def test_empty_inputs_and_zero_rays():
    mesh = scenes.torus(100)
    fr = kernel.driver_frame(0)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    hit, tuv, point = sc.trace_rays(np.zeros((0, 3)), np.zeros((0, 3)))
    assert hit.numel() == 0
    assert sc.occluded(np.zeros((0, 3)), np.zeros((0, 3))).numel() == 0
    # zero-length rays and unused (NaN-origin) slots hit nothing
    S = np.array([[0.0, 0, 0], [np.nan, 0, 0], [1.0, 2, 3]])
    E = np.array([[0.0, 0, 0], [0.0, 0, 1], [1.0, 2, 3]])
    hit, _, _ = sc.trace_rays(S, E)
    assert hit.cpu().tolist() == [-1, -1, -1]
    assert sc.occluded(S, E).cpu().tolist() == [0, 0, 0]
    sc.close()
    # a scene with zero triangles renders the miss colour everywhere (driver.c:136-139)
    empty = scenes.torus(100)
    for k in ("A", "B", "C", "An", "Bn", "Cn"):
        setattr(empty, k, np.zeros(0, np.int32))
    sc2 = kernel.Scene(empty)
    sc2.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc2.render_rows(fr, 20, 10, TAN_HALF)
    assert torch.all(out["hit"] == -1) and torch.all(out["rgb8"] == 127)
    sc2.close()


def test_degenerate_and_duplicate_triangles():
    """Zero-area triangles never hit (den == 0, triangle_intersection.cu:67); exact duplicates
    tie and the lowest index wins (:169)."""
    base = scenes.torus(200)
    m = base.copy()
    n = m.num_tris
    # append: 10 degenerate (A == B) and a duplicate of every 7th triangle, BEFORE and AFTER
    dup = np.arange(0, n, 7, dtype=np.int32)
    def cat(a, extra_front, extra_back):
        return np.ascontiguousarray(np.concatenate([extra_front, a, extra_back]).astype(np.int32))
    deg = m.A[:10]
    m.A, m.B, m.C = cat(m.A, m.A[dup], deg), cat(m.B, m.B[dup], deg), cat(m.C, m.C[dup], m.C[:10])
    m.An, m.Bn, m.Cn = cat(m.An, m.An[dup], m.An[:10]), cat(m.Bn, m.Bn[dup], m.Bn[:10]), cat(m.Cn, m.Cn[dup], m.Cn[:10])
    fr = kernel.driver_frame(6)
    W, H = 96, 72
    ref, _ = _ref(m, fr, W, H)
    sc, out = _gpu(m, fr, W, H)
    assert np.array_equal(out["hit"], ref["hit"])
    hits = ref["hit"][ref["hit"] >= 0]
    assert hits.max() < len(dup) + n                 # never a degenerate one
    assert (hits < len(dup)).any()                   # front duplicates beat their originals
    sc.close()


def test_candidate_list_overflow_falls_back_to_full_fp64_sweep():
    """40 coincident triangles: more candidates than list slots -> the ray is re-swept in fp64
    (closest hit) / resolved by the full any-hit sweep, still with the reference's answer."""
    pts = np.array([[0, 0, 0], [-3, -3, 12.0], [5, -3, 12.0], [-3, 5, 12.0],
                    [-3, -3, 20.0], [5, -3, 20.0], [-3, 5, 20.0]], float)
    k = 40
    A = np.array([4] * 3 + [1] * k, np.int32)        # three far copies first, then the near stack
    B = np.array([5] * 3 + [2] * k, np.int32)
    Cc = np.array([6] * 3 + [3] * k, np.int32)
    nrm = np.array([[0, 0, 0], [0, 0, -1.0]])
    one = np.ones(len(A), np.int32)
    mesh = scenes.Mesh(x=pts[:, 0].copy(), y=pts[:, 1].copy(), z=pts[:, 2].copy(),
                       xn=nrm[:, 0].copy(), yn=nrm[:, 1].copy(), zn=nrm[:, 2].copy(),
                       A=A, B=B, C=Cc, An=one, Bn=one.copy(), Cn=one.copy())
    sc = kernel.Scene(mesh)
    sc.transform(np.eye(4))
    osc = O.Scene(mesh)
    rng = np.random.default_rng(0)
    S = np.zeros((64, 3))
    E = np.stack([rng.uniform(-0.1, 0.1, 64), rng.uniform(-0.1, 0.1, 64), np.ones(64)], 1)
    kernel.profile_enable(True)
    kernel.profile_reset()
    hit, tuv, _ = sc.trace_rays(S, E)
    occ = sc.occluded(S, E)
    prof = kernel.profile_get()
    kernel.profile_enable(False)
    assert prof.exact_overflow >= 32                 # the fallback really ran for most rays
    nhit = 0
    for i in range(64):
        idx, t, uv, _ = O.intersect(osc, S[i], E[i])
        assert hit[i].item() == idx and idx in (3, -1)   # 3 = lowest index of the nearest stack
        assert occ[i].item() == (1 if idx != -1 else 0)
        if idx != -1:
            nhit += 1
            assert tuv[i, 0].item() == t
    assert nhit >= 32
    sc.close()


def test_textured_materials():
    """map_Ka/Kd/Ks lookups (graphic_tools.c:161-197): nearest texel, v flipped."""
    rng = np.random.default_rng(5)
    mesh = scenes.torus(400, with_uv=True)
    tex0 = rng.integers(0, 256, size=(16, 32, 3), dtype=np.uint8)
    tex1 = rng.integers(0, 256, size=(9, 5, 3), dtype=np.uint8)
    mesh.textures = [tex0, tex1]
    mesh.Ka = np.array([[0.2, 0.3, 0.4], [1.0, 1.0, 1.0]])
    mesh.Kd = np.array([[0.9, 0.8, 0.7], [0.5, 0.5, 0.5]])
    mesh.Ks = np.array([[0.1, 0.2, 0.3], [1.0, 0.0, 1.0]])
    mesh.map_Ka = np.array([-1, 1], np.int32)
    mesh.map_Kd = np.array([0, -1], np.int32)
    mesh.map_Ks = np.array([-1, 0], np.int32)
    mesh.mtl = (np.arange(mesh.num_tris) % 3 - 1).astype(np.int32)     # -1, 0, 1, ...
    fr = kernel.driver_frame(8)
    W, H = 96, 72
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert np.array_equal(out["hit"], ref["hit"])
    assert np.abs(out["argb"] - ref["argb"]).max() < 1e-9
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    used = set(mesh.mtl[ref["hit"][ref["hit"] >= 0]].tolist())
    assert used == {-1, 0, 1}
    sc.close()


@pytest.mark.parametrize("nl", [0, 1, 5, 11])      # 11: more lights than one any-hit launch takes
def test_light_counts(nl):
    mesh = scenes.sphere(300)
    fr = kernel.driver_frame(3)
    rng = np.random.default_rng(nl)
    fr.num_lights = nl
    for i in range(nl):
        pos = rng.uniform(-25, 25, 3)
        for k in range(3):
            fr.light_in_eye_space[i][k] = pos[k] + (0, 0, 18)[k]
            fr.light_rgb[i][k] = rng.uniform(0, 1)
        fr.light_attenuation[i][0], fr.light_attenuation[i][1] = 0.05, 0.01
        fr.light_max_dist[i] = (30.0, 100.0)[i % 2]          # some beyond light_max_dist
    W, H = 64, 48
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert np.array_equal(out["hit"], ref["hit"])
    assert np.abs(out["argb"] - ref["argb"]).max() < 1e-9
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    if nl == 0:
        assert ref["shadow_rays"] == 0
    sc.close()


def test_error_codes_instead_of_faults():
    lib = _lib.load()
    mesh = scenes.torus(50)
    sc = kernel.Scene(mesh)
    hit = torch.empty(16, dtype=torch.int32, device="cuda")
    # tracing before any transform
    assert lib.rt_trace_primary(sc.h, 4, 4, TAN_HALF, 0, 4, hit.data_ptr(), None) == -3
    assert b"rt_scene_transform" in lib.rt_last_error()
    sc.transform(np.eye(4))
    assert lib.rt_trace_primary(sc.h, 4, 4, TAN_HALF, 3, 2, hit.data_ptr(), None) == -1      # empty band
    assert lib.rt_trace_primary(sc.h, 4, 4, TAN_HALF, 0, 5, hit.data_ptr(), None) == -1      # band past H
    assert lib.rt_trace_primary(sc.h, 4, 4, -1.0, 0, 4, hit.data_ptr(), None) == -1
    assert lib.rt_trace_primary(sc.h, 4, 4, TAN_HALF, 0, 4, None, None) == -1
    assert lib.rt_trace_rays(sc.h, -1, None, None, None, None, None) == -1
    # out-of-range indices are rejected at upload, never seen by a kernel
    bad = mesh.A.copy()
    bad[3] = mesh.num_v + 5
    rc = lib.rt_scene_set_triangles(sc.h, mesh.num_tris, bad.ctypes.data_as(_lib.c_int_p),
                                    mesh.B.ctypes.data_as(_lib.c_int_p), mesh.C.ctypes.data_as(_lib.c_int_p),
                                    None, None, None, None, None, None, None)
    assert rc == -1 and b"vertex index" in lib.rt_last_error()
    fr = kernel.driver_frame(0)
    fr.num_lights = 101
    assert lib.rt_render_rows(sc.h, C.byref(fr), 4, 4, TAN_HALF, 0, 4, hit.data_ptr(), None, None, None) == -1
    assert lib.rt_init(99) == -1
    # row sets
    fr.num_lights = 2
    for bad in ((0, 4, 0, 2, 0), (0, 4, 2, 0, 0), (0, 4, 2, 2, 2), (0, 5, 2, 2, 0), (-1, 4, 2, 2, 0), (3, 2, 2, 2, 0)):
        rows = _lib.RtRows(*bad)
        assert lib.rt_render_rows_part(sc.h, C.byref(fr), 4, 4, TAN_HALF, C.byref(rows), hit.data_ptr(),
                                       None, None, None) == -1, bad
        assert lib.rt_place_rows_part(hit.data_ptr(), 4, 4, C.byref(rows), hit.data_ptr()) == -1, bad
    sc.close()


def test_inconsistent_material_tables_are_errors_not_faults(tmp_path):
    """Index tables that do not fit each other never reach the shading kernel: material and
    texture numbers out of range and textured materials on triangles without texture indices
    are refused with a status code."""
    lib = _lib.load()
    fr = kernel.driver_frame(0)
    hit = torch.empty(64 * 48, dtype=torch.int32, device="cuda")

    def render(sc):
        return lib.rt_render_rows(sc.h, C.byref(fr), 64, 48, TAN_HALF, 0, 48, hit.data_ptr(), None, None, None)

    # 1. a textured material, triangles without At/Bt/Ct
    mesh = scenes.with_materials(scenes.torus(200), num_mats=2)
    mesh.textures = [np.full((4, 4, 3), 200, np.uint8)]
    mesh.map_Kd = np.array([0, -1], np.int32)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    assert render(sc) == -3 and b"texture coordinates" in lib.rt_last_error()
    sc.close()
    # 2. a texture number beyond the textures that were set
    mesh = scenes.with_materials(scenes.torus(200, with_uv=True), num_mats=2)
    mesh.textures = [np.full((4, 4, 3), 200, np.uint8)]
    mesh.map_Ks = np.array([-1, 1], np.int32)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    assert render(sc) == -3 and b"texture 1" in lib.rt_last_error()
    sc.close()
    # 3. a material index beyond the material table; below -1; vertex slot 0
    mesh = scenes.with_materials(scenes.torus(200), num_mats=2)
    mesh.mtl = mesh.mtl.copy()
    mesh.mtl[7] = 2
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    assert render(sc) == -3 and b"material 2" in lib.rt_last_error()
    ip = lambda a: a.ctypes.data_as(_lib.c_int_p)
    low = mesh.mtl.copy()
    low[0] = -2
    args = [ip(mesh.A), ip(mesh.B), ip(mesh.C), ip(mesh.An), ip(mesh.Bn), ip(mesh.Cn), None, None, None]
    assert lib.rt_scene_set_triangles(sc.h, mesh.num_tris, *args, ip(low)) == -1
    zero = mesh.B.copy()
    zero[5] = 0
    args[1] = ip(zero)
    assert lib.rt_scene_set_triangles(sc.h, mesh.num_tris, *args, ip(mesh.mtl)) == -1
    assert b"[1,num_v]" in lib.rt_last_error()
    sc.close()
    # 4. the loader: a material with map_Kd on faces without vt (v//vn) is shaded with the
    #    material's constant colours, like a texture that failed to load
    ppm = tmp_path / "t.ppm"
    ppm.write_bytes(b"P6\n2 2\n255\n" + bytes([255, 0, 0] * 4))
    images = []
    for name, map_line in (("with_map", "map_Kd t.ppm\nmap_Ka t.ppm\n"), ("without", "")):
        (tmp_path / f"{name}.mtl").write_text(
            "newmtl red\nKa 0.3 0.3 0.3\nKd 0.2 0.9 0.4\nKs 0.5 0.5 0.5\n" + map_line)
        (tmp_path / f"{name}.obj").write_text(
            f"mtllib {name}.mtl\nv -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nvn 0 0 1\nusemtl red\n"
            "f 1//1 2//1 3//1 4//1\n")
        h = C.c_void_p()
        _lib.check(lib.rt_obj_load(str(tmp_path / f"{name}.obj").encode(), C.byref(h)), "rt_obj_load")
        sc = kernel.Scene()
        _lib.check(lib.rt_scene_set_obj(sc.h, h), "rt_scene_set_obj")
        sc.transform(kernel.frame_matrix(fr, "obmat"))
        argb = torch.empty((48, 64, 3), dtype=torch.float64, device="cuda")
        _lib.check(lib.rt_render_rows(sc.h, C.byref(fr), 64, 48, TAN_HALF, 0, 48, hit.data_ptr(), None,
                                      argb.data_ptr(), None), "rt_render_rows")
        assert int((hit.view(48, 64) >= 0).sum()) > 100
        images.append(argb)
        lib.rt_obj_free(h)
        sc.close()
    assert torch.equal(images[0], images[1])


@pytest.mark.parametrize("dist", [300.0, 3.0e5])
def test_far_ray_origins(dist):
    """The general-position sweep works with the ray's moment (a property of its line), so the
    filter's rounding bound does not depend on how far away the ray starts; lines that pass
    the mesh by more than its extent are retired at set-up."""
    mesh = scenes.sphere(400)
    fr = kernel.driver_frame(0)
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    osc = O.Scene(mesh, X, Y, Z)
    c = np.array([X[1:].mean(), Y[1:].mean(), Z[1:].mean()])
    rng = np.random.default_rng(2)
    d = rng.normal(size=(500, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    S = c + dist * d
    E = c + rng.uniform(-6, 6, size=(500, 3))
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    hit, _, _ = sc.trace_rays(S, E)
    hit = hit.cpu().numpy()
    assert sum(int(hit[k] != O.intersect(osc, S[k], E[k])[0]) for k in range(500)) == 0
    sc.close()


def _eye_space_mesh(tris):
    """Mesh from a list of triangles given directly in eye space (used with an identity obmat)."""
    P = np.vstack([np.zeros((1, 3))] + [np.asarray(t, float) for t in tris])
    n = len(tris)
    A = np.arange(1, 3 * n + 1, 3, dtype=np.int32)
    nrm = np.vstack([np.zeros((1, 3))] + [np.cross(np.subtract(t[1], t[0]), np.subtract(t[2], t[0]))[None] for t in tris])
    idn = np.arange(1, n + 1, dtype=np.int32)
    return scenes.Mesh(x=P[:, 0].copy(), y=P[:, 1].copy(), z=P[:, 2].copy(), xn=nrm[:, 0].copy(),
                       yn=nrm[:, 1].copy(), zn=nrm[:, 2].copy(), A=A, B=A + 1, C=A + 2,
                       An=idn, Bn=idn.copy(), Cn=idn.copy())


@pytest.mark.parametrize("through_eye", [3, 40])     # 40: more than a candidate list holds
def test_triangles_whose_plane_contains_the_eye(through_eye):
    """The pixel-grid sweep decides a pair from sign bits after orienting every triangle towards
    the eye (t > 0 fixes the sign of the determinant, triangle_intersection.cu:59-94).  A
    triangle seen exactly or almost edge-on has no orientation: it must be left to the fp64
    pass, together with triangles behind the eye and across the eye plane, without disturbing
    the ordinary ones."""
    rng = np.random.default_rng(through_eye)
    tris = []
    for k in range(30):             # ordinary triangles in front of the eye, both windings
        c = np.array([rng.uniform(-6, 6), rng.uniform(-4, 4), rng.uniform(10, 30)])
        e1, e2 = rng.normal(size=3) * 2, rng.normal(size=3) * 2
        tris.append([c, c + e1, c + e2] if k % 2 else [c, c + e2, c + e1])
    for k in range(through_eye):    # planes through the eye (the origin of eye space), or 1e-13 off it
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        b1 = np.cross(n, [0.3, 0.5, 0.8]); b1 /= np.linalg.norm(b1)
        b2 = np.cross(n, b1)
        off = n * (0.0, 1e-13, 1e-9, 1e-7, 1e-5, 1e-3)[k % 6]       # through the eye, or this far off it
        c = b1 * rng.uniform(-8, 8) + b2 * rng.uniform(-8, 8)
        if k % 4 == 0:
            c = c * 0.01            # the eye inside or next to the triangle
        tris.append([c + off, c + 3 * b1 + off, c + 3 * b2 + off])
    for k in range(6):              # behind the eye, and across the eye plane z = 0
        c = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), -5.0 if k < 3 else 0.0])
        tris.append([c + [0, 0, -1], c + [2, 0, 1.5], c + [0, 2, 1.5]])
    order = rng.permutation(len(tris))
    mesh = _eye_space_mesh([tris[i] for i in order])
    fr = kernel.driver_frame(0)
    for r in range(4):
        for c in range(4):
            fr.obmat[r][c] = fr.obinv[r][c] = 1.0 if r == c else 0.0
    W, H = 96, 64
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert (ref["hit"] >= 0).mean() > 0.05
    assert np.array_equal(out["hit"], ref["hit"])
    m = ref["hit"] >= 0
    assert np.array_equal(out["tuv"][m], ref["tuv"][m])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    sc.close()


def test_triangles_whose_plane_contains_a_light():
    """The shadow rays' first test bounds a triangle's projection through the light; a triangle
    whose plane passes through the light (or within rounding of it), one that surrounds it, and
    ones that straddle the planes of the three axes through it have no such bound and must be
    left to the later tests without disturbing the ordinary ones."""
    rng = np.random.default_rng(7)
    L = np.array([1.5, 2.0, 14.0])
    tris = []
    for k in range(40):             # a floor and clutter that receive shadows
        c = np.array([rng.uniform(-7, 7), rng.uniform(-5, 5), rng.uniform(16, 26)])
        e1, e2 = rng.normal(size=3) * 2.5, rng.normal(size=3) * 2.5
        tris.append([c, c + e1, c + e2])
    for k in range(24):             # planes through the light, or a little off it
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        b1 = np.cross(n, [0.3, 0.5, 0.8]); b1 /= np.linalg.norm(b1)
        b2 = np.cross(n, b1)
        off = n * (0.0, 1e-13, 1e-9, 1e-7, 1e-5, 1e-3)[k % 6]
        c = L + b1 * rng.uniform(-3, 3) + b2 * rng.uniform(-3, 3)
        if k % 4 == 0:
            c = L + (c - L) * 0.01  # the light inside or next to the triangle
        tris.append([c + off, c + 2 * b1 + off, c + 2 * b2 + off])
    for k in range(9):              # across the planes x = Lx, y = Ly, z = Lz, near the light and far from it
        ax = k % 3
        c = L + rng.normal(size=3) * (0.3 if k < 3 else 4.0)
        c[ax] = L[ax]
        e1, e2 = rng.normal(size=3), rng.normal(size=3)
        tris.append([c - 0.5 * e1, c + e1, c + e2])
    order = rng.permutation(len(tris))
    mesh = _eye_space_mesh([tris[i] for i in order])
    fr = kernel.driver_frame(0)
    for r in range(4):
        for c in range(4):
            fr.obmat[r][c] = fr.obinv[r][c] = 1.0 if r == c else 0.0
    fr.num_lights = 2
    for c in range(3):
        fr.light_in_eye_space[0][c] = L[c]
        fr.light_in_eye_space[1][c] = (-6.0, 7.0, 3.0)[c]
    W, H = 128, 80
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert (ref["hit"] >= 0).mean() > 0.2 and ref["shadow_rays"] > 2000
    assert np.array_equal(out["hit"], ref["hit"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    # ... and the same frame with the ranged, sorted and culled paths
    for env in ({"RT_AMD_RANGE_MIN_WORK": "0"}, {"RT_AMD_RANGE_MIN_WORK": "0", "RT_AMD_CULL": "1"}):
        import os
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            again = {k: v.cpu().numpy() for k, v in sc.render_rows(fr, W, H, TAN_HALF).items()}
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        assert np.array_equal(again["hit"], out["hit"]) and np.array_equal(again["rgb8"], out["rgb8"]), env
    sc.close()


@pytest.mark.parametrize("shape", ["16,1", "16,3", "8,2", "4,1", "2,5", "1,1", "16,2,2", "16,1,1", "8,3,1", "4,2,2",
                                   "32,1", "32,2,2", "32,3,1", "64,1", "64,2,1", "64,3,4", "128,1", "128,3,2", "128,2,1",
                                   "256,1", "256,2,4", "256,5,1", "512,1", "512,3,0", "1024,1", "1024,2,0", "1024,4,1", "2048,1", "2048,3,0", "2048,2,1",
                                   "4096,1", "4096,3,0", "4096,2,1", "4096,3,-1", "2048,2,-1", "256,2,-1", "16,3,-1", "1,2,-1",
                                   "8192,1", "8192,2,-1", "8192,3,1", "16384,1", "16384,3,-1", "16384,2,0",
                                   "32768,2", "65536,1", "65536,5,-1", "147456,1", "147456,3,-1"])
@pytest.mark.parametrize("W,H", [(500, 20), (37, 19), (200, 150), (90, 600)])
@pytest.mark.parametrize("regular", [None, "0"])
def test_every_shape_of_the_pixel_grid_sweep(shape, W, H, monkeypatch, regular):
    """The launcher picks pixels per lane and triangle chunks from the image and mesh size, so
    small inputs never reach the shapes large frames run with: force each one (lanes whose
    pixels are partly or wholly off the image, chunks of uneven length, the four waves of a
    workgroup side by side, two by two or stacked) and compare with the oracle.  Each shape in
    both forms of the kernel: rows stepped by a constant (the default where the rows of the
    call are regular, and the only one with blocks of 64 to 2304 rows -- the wave-block form, whose
    workgroups always own one block each), and a coordinate per row.  The 600-row image gives the tall
    blocks several row groups and a second block of rows."""
    if regular == "0" and int(shape.split(",")[0]) >= 4096:
        pytest.skip("blocks of 64 rows and more exist in the regular-row form only")
    if H == 600 and int(shape.split(",")[0]) < 1024:
        pytest.skip("the tall image is for the tall lanes")
    if regular is not None:
        monkeypatch.setenv("RT_AMD_RASTER_REGULAR", regular)
    monkeypatch.setenv("RT_AMD_RASTER_FORCE", shape)
    mesh = scenes.triangle_soup(700, seed=5)
    fr = kernel.driver_frame(4)
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert (ref["hit"] >= 0).mean() > 0.05
    assert np.array_equal(out["hit"], ref["hit"])
    m = ref["hit"] >= 0
    assert np.array_equal(out["tuv"][m], ref["tuv"][m])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    sc.close()


@pytest.mark.parametrize("shape", ["16,1", "16,3", "8,1", "8,4", "4,3", "2,2", "1,7"])
def test_every_shape_of_the_any_hit_sweep(shape, monkeypatch):
    """Same for the shadow rays' sweep: rays per lane and triangle chunks that share a ray's
    candidate list through atomics (more chunks than a few-tile range can feed included)."""
    monkeypatch.setenv("RT_AMD_APEX_FORCE", shape)
    mesh = scenes.triangle_soup(900, seed=9)
    fr = kernel.driver_frame(6)
    W, H = 150, 70
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert ref["shadow_rays"] > 1000
    assert np.array_equal(out["hit"], ref["hit"])
    assert np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int)).max() <= 1
    sc.close()


@pytest.mark.parametrize("apex_shape", [None, "8,3", "2,5"])
def test_ranged_any_hit_sweeps_on_a_small_scene(apex_shape, monkeypatch):
    """Large frames sweep the shadow rays in triangle ranges with the survivors compacted (and
    sorted by direction) in between; small inputs take one launch.  Force the ranged path on a
    scene the oracle can shade completely: 10 000 triangles = 79 tiles = ranges of 16, 32 and 31
    tiles, four lights (one inside the mesh, one far away)."""
    monkeypatch.setenv("RT_AMD_RANGE_MIN_WORK", "0")
    if apex_shape:
        monkeypatch.setenv("RT_AMD_APEX_FORCE", apex_shape)
    mesh = scenes.triangle_soup(10000, seed=21)
    fr = kernel.driver_frame(9)
    for k, pos in enumerate([(0.5, 0.2, 12.0), (300.0, -200.0, 150.0)]):
        l = fr.num_lights
        for c in range(3):
            fr.light_in_eye_space[l][c] = pos[c]
            fr.light_rgb[l][c] = 0.5
        fr.light_attenuation[l][0] = fr.light_attenuation[l][1] = 0.0
        fr.light_max_dist[l] = 1e9
        fr.num_lights = l + 1
    W, H = 160, 90
    ref, _ = _ref(mesh, fr, W, H)
    sc, out = _gpu(mesh, fr, W, H)
    assert ref["shadow_rays"] > 20000
    assert np.array_equal(out["hit"], ref["hit"])
    d = np.abs(out["rgb8"].astype(int) - ref["rgb8"].astype(int))
    assert d.max() <= 1
    sc.close()
