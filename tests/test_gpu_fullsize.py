"""BASELINE.json configurations at full size on the GPU.

The CPU oracle cannot sweep a whole 4K frame of a large mesh in seconds, so each case
compares (a) every pixel of a deterministic row sample against the oracle, bit-exact, and
(b) size-independent properties over the full frame: band-split invariance, determinism
(two renders identical), index/barycentric ranges, and t/u/v recomputed independently in
numpy fp64 for every hit pixel from the winning triangle alone.
Named meshes (teapot, bunny, ...) are not available offline; procedural meshes of the same
triangle counts stand in (see bench.py WORKLOADS).
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from raytracer_amd import kernel, scenes
from conftest import TAN_HALF

pytestmark = pytest.mark.gpu


def _recompute_tuv(mesh, X, Y, Z, hit, W, H):
    """Independent numpy restatement of triangle_intersection.cu:59-86 for the winning
    triangle of every hit pixel (vectorised; no contraction in numpy)."""
    ys, xs = np.nonzero(hit >= 0)
    tri = hit[ys, xs]
    A = np.stack([X[mesh.A[tri]], Y[mesh.A[tri]], Z[mesh.A[tri]]], 1)
    B = np.stack([X[mesh.B[tri]], Y[mesh.B[tri]], Z[mesh.B[tri]]], 1)
    Cc = np.stack([X[mesh.C[tri]], Y[mesh.C[tri]], Z[mesh.C[tri]]], 1)
    S = np.zeros_like(A)
    E = np.stack([(xs - W // 2).astype(float), (ys - H // 2).astype(float),
                  np.full(len(xs), (W // 2) / TAN_HALF)], 1)
    AB, AC, ES, AS = B - A, Cc - A, S - E, S - A

    def det(a, b, c):
        d = 0.0 + a[:, 0] * (b[:, 1] * c[:, 2] - b[:, 2] * c[:, 1])
        d = d + (-b[:, 0]) * (a[:, 1] * c[:, 2] - a[:, 2] * c[:, 1])
        return d + c[:, 0] * (a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1])

    den = det(AB, AC, ES)
    return ys, xs, det(AB, AC, AS) / den, det(AS, AC, ES) / den, det(AB, AS, ES) / den


def _check_case(mesh, frame_no, W, H, sample_rows, shade_rows=0, bands=3):
    fr = kernel.driver_frame(frame_no)
    obmat = kernel.frame_matrix(fr, "obmat")
    sc = kernel.Scene(mesh)
    sc.transform(obmat)
    hit_t, tuv_t = sc.trace_primary(W, H, TAN_HALF)
    hit, tuv = hit_t.cpu().numpy(), tuv_t.cpu().numpy()
    nhit = int((hit >= 0).sum())
    assert nhit > 0.02 * W * H, "mesh does not cover enough of the frame"
    # ranges
    assert hit.min() >= -1 and hit.max() < mesh.num_tris
    m = hit >= 0
    assert np.all(tuv[m][:, 0] > 0)
    assert np.all((tuv[m][:, 1] >= 0) & (tuv[m][:, 2] >= 0) & (tuv[m][:, 1] + tuv[m][:, 2] <= 1))
    # determinism
    hit2, tuv2 = sc.trace_primary(W, H, TAN_HALF)
    assert torch.equal(hit2, hit_t) and torch.equal(tuv2[hit_t >= 0], tuv_t[hit_t >= 0])
    # band-split invariance (what the multi-GPU sharding relies on)
    parts = [sc.trace_primary(W, H, TAN_HALF, *kernel.band(H, bands, r))[0] for r in range(bands)]
    assert torch.equal(torch.cat(parts, 0), hit_t)
    # independent fp64 recomputation of t,u,v for every hit pixel
    X, Y, Z = sc.eye_vertices()
    ys, xs, t, u, v = _recompute_tuv(mesh, X, Y, Z, hit, W, H)
    assert np.array_equal(tuv[ys, xs, 0], t) and np.array_equal(tuv[ys, xs, 1], u) and np.array_equal(tuv[ys, xs, 2], v)
    # oracle on a row sample: every pixel of those rows, bit-exact
    XO, YO, ZO = O.transform(mesh, obmat)
    assert np.array_equal(XO, X) and np.array_equal(YO, Y) and np.array_equal(ZO, Z)
    osc = O.Scene(mesh, XO, YO, ZO)
    step = max(H // sample_rows, 1)
    y0 = step // 2
    ref = O.render_rows(osc, O.frame_from(fr), W, H, TAN_HALF, y0=y0, y1=H, ystep=step, shade=False)
    rows = np.arange(y0, H, step)
    assert np.array_equal(hit[rows], ref["hit"][rows]), "hit index differs from the oracle"
    mr = ref["hit"][rows] >= 0
    assert np.array_equal(tuv[rows][mr], ref["tuv"][rows][mr])
    mism = int((hit[rows] != ref["hit"][rows]).sum())
    # shaded rows: RGB within 1/255
    if shade_rows:
        sstep = max(H // shade_rows, 1)
        srows = np.arange(sstep // 2, H, sstep)
        for y in srows[:shade_rows]:
            out = sc.render_rows(fr, W, H, TAN_HALF, int(y), int(y) + 1, want=("hit", "rgb8"))
            r1 = O.render_rows(osc, O.frame_from(fr), W, H, TAN_HALF, y0=int(y), y1=int(y) + 1)
            assert np.array_equal(out["hit"].cpu().numpy()[0], r1["hit"][y])
            d = np.abs(out["rgb8"].cpu().numpy()[0].astype(int) - r1["rgb8"][y].astype(int))
            assert d.max() <= 1
    sc.close()
    return nhit, len(rows) * W, mism


def test_config1_teapot_class_640x480_full_frame():
    """Config 1: ~6.3k triangles with quads split, 640x480: the WHOLE frame against the
    oracle, hit ids bit-exact and RGB within 1/255."""
    mesh = scenes.torus(6320)
    fr = kernel.driver_frame(0)
    W, H = 640, 480
    obmat = kernel.frame_matrix(fr, "obmat")
    sc = kernel.Scene(mesh)
    sc.transform(obmat)
    out = sc.render_rows(fr, W, H, TAN_HALF)
    X, Y, Z = O.transform(mesh, obmat)
    ref = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF)
    hit = out["hit"].cpu().numpy()
    assert (ref["hit"] >= 0).sum() > 20000
    assert np.array_equal(hit, ref["hit"])              # 100 % of 307 200 pixels
    m = hit >= 0
    assert np.array_equal(out["tuv"].cpu().numpy()[m], ref["tuv"][m])
    d = np.abs(out["rgb8"].cpu().numpy().astype(int) - ref["rgb8"].astype(int))
    assert d.max() <= 1
    assert (d > 0).mean() < 1e-3                         # and nearly all channels identical
    sc.close()


def test_config2_mesh_1080p():
    nhit, checked, mism = _check_case(scenes.sphere(20000), 3, 1920, 1080, sample_rows=12, shade_rows=1)
    assert mism == 0 and checked >= 12 * 1920


def test_config3_bunny_class_4k():
    nhit, checked, mism = _check_case(scenes.torus(69451), 7, 3840, 2160, sample_rows=6, shade_rows=1)
    assert mism == 0


def test_config4_soup_1m_4k_rows():
    """Config 4 geometry (1M-triangle soup, 3840x2160): two full rows against the oracle and
    band/determinism properties on a 128-row band (primary rays only; the shaded path is
    test_config4_soup_1m_4k_shaded_band)."""
    mesh = scenes.triangle_soup(1_000_000, seed=12345)
    fr = kernel.driver_frame(0)
    obmat = kernel.frame_matrix(fr, "obmat")
    W, H = 3840, 2160
    sc = kernel.Scene(mesh)
    sc.transform(obmat)
    y0, y1 = 1000, 1128
    hit_t, tuv_t = sc.trace_primary(W, H, TAN_HALF, y0, y1)
    a, _ = sc.trace_primary(W, H, TAN_HALF, y0, y0 + 50)
    b, _ = sc.trace_primary(W, H, TAN_HALF, y0 + 50, y1)
    assert torch.equal(torch.cat([a, b], 0), hit_t)
    hit = hit_t.cpu().numpy()
    assert (hit >= 0).mean() > 0.9
    X, Y, Z = O.transform(mesh, obmat)
    osc = O.Scene(mesh, X, Y, Z)
    for y in (1003, 1100):
        ref = O.render_rows(osc, O.frame_from(fr), W, H, TAN_HALF, y0=y, y1=y + 1, shade=False)
        assert np.array_equal(hit[y - y0], ref["hit"][y])
        m = ref["hit"][y] >= 0
        assert np.array_equal(tuv_t.cpu().numpy()[y - y0][m], ref["tuv"][y][m])
    sc.close()


def _shaded_rows_vs_oracle(mesh, frame_no, W, H, band, check_rows, min_shadow_launches, min_cell_launches=0):
    """The whole pixel loop on a band tall enough that the shadow phase takes its production
    route by itself (default thresholds): occluder cache, then either the sweep over lattices of
    direction cells (min_cell_launches) or, with that switched off, rays sorted by direction and
    triangle ranges with survivor compaction.  Full rows of the band against the oracle's shaded rows."""
    fr = kernel.driver_frame(frame_no)
    obmat = kernel.frame_matrix(fr, "obmat")
    sc = kernel.Scene(mesh)
    sc.transform(obmat)
    y0, y1 = band
    kernel.profile_enable(True)
    kernel.profile_reset()
    out = sc.render_rows(fr, W, H, TAN_HALF, y0, y1)
    prof = kernel.profile_get()
    kernel.profile_enable(False)
    # several any-hit launches per cache level = the ranged branch ran (a small sweep is one launch);
    # one cell-lattice launch per cache level = the lattice ran
    assert prof.shadow_launches >= min_shadow_launches, prof.shadow_launches
    assert prof.cell_launches >= min_cell_launches, prof.cell_launches
    if min_cell_launches == 0:
        assert prof.cell_launches == 0
    assert prof.rays_shadow > 0 and prof.shadow_tests > 0
    # the same band again without the profiling syncs in between: identical
    again = sc.render_rows(fr, W, H, TAN_HALF, y0, y1, want=("hit", "rgb8"))
    assert torch.equal(again["hit"], out["hit"]) and torch.equal(again["rgb8"], out["rgb8"])
    hit, tuv = out["hit"].cpu().numpy(), out["tuv"].cpu().numpy()
    argb, rgb8 = out["argb"].cpu().numpy(), out["rgb8"].cpu().numpy()
    X, Y, Z = O.transform(mesh, obmat)
    osc = O.Scene(mesh, X, Y, Z)
    shadowed = lit = 0
    for y in check_rows:
        ref = O.render_rows(osc, O.frame_from(fr), W, H, TAN_HALF, y0=y, y1=y + 1, shade=True)
        k = y - y0
        assert np.array_equal(hit[k], ref["hit"][y]), f"row {y}: hit index differs from the oracle"
        m = ref["hit"][y] >= 0
        assert m.sum() > 0.02 * W
        assert np.array_equal(tuv[k][m], ref["tuv"][y][m])
        assert np.abs(argb[k] - ref["argb"][y]).max() < 1e-9          # libm pow vs the device's
        assert np.abs(rgb8[k].astype(int) - ref["rgb8"][y].astype(int)).max() <= 1   # 1/255 per channel
        assert ref["shadow_rays"] > 0
        # pixels no light reaches (occluded, facing away or out of range) show only the ambient
        # term (light_model.c:133-139 with K = 1); the rows must hold both kinds
        only_ambient = np.abs(ref["argb"][y][m] - 0.2).max(axis=1) < 1e-12
        shadowed += int(only_ambient.sum())
        lit += int((~only_ambient).sum())
    assert shadowed > 0 and lit > 0
    sc.close()


def test_config4_soup_1m_4k_shaded_band():
    """Config 4 through the whole pixel loop at production scale: 1 M triangles, 3840 wide, a
    64-row band (2.5e11 primary pairs; 3e4 + 4e5 shadow rays x 1 M triangles, far above the
    1.5e10-pair threshold of the ranged sweeps), two full rows against the oracle
    (light_model.c:60-76: each hit pixel's shadow rays against every triangle)."""
    _shaded_rows_vs_oracle(scenes.triangle_soup(1_000_000, seed=12345), 0, 3840, 2160,
                           band=(1000, 1064), check_rows=(1003, 1050), min_shadow_launches=2, min_cell_launches=2)


def test_config4_soup_1m_4k_shaded_band_ranged_sweeps(monkeypatch):
    """The same band with the cell lattice switched off: the ray-by-ray any-hit sweep in triangle
    ranges with survivor compaction, as every phase too small for the lattice runs."""
    monkeypatch.setenv("RT_AMD_SHADOW_LATTICE", "0")
    _shaded_rows_vs_oracle(scenes.triangle_soup(1_000_000, seed=12345), 0, 3840, 2160,
                           band=(1000, 1064), check_rows=(1003, 1050), min_shadow_launches=8)


def test_config3_bunny_class_4k_shaded_frame():
    """Config 3 through the whole pixel loop: the full 3840x2160 frame of the 69 451-triangle
    mesh (its shadow rays x triangles cross the ranged sweeps' threshold), six full rows against
    the oracle."""
    _shaded_rows_vs_oracle(scenes.torus(69451), 7, 3840, 2160, band=(0, 2160),
                           check_rows=(200, 700, 1000, 1081, 1400, 1900), min_shadow_launches=2, min_cell_launches=1)


def test_config3_bunny_class_4k_shaded_frame_ranged_sweeps(monkeypatch):
    monkeypatch.setenv("RT_AMD_SHADOW_LATTICE", "0")
    _shaded_rows_vs_oracle(scenes.torus(69451), 7, 3840, 2160, band=(0, 2160),
                           check_rows=(200, 1081, 1900), min_shadow_launches=5)


def _golden(name):
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz has not been generated (tests/golden/make_fullframe_golden.py)")
    return np.load(path)


def test_config2_whole_1080p_frame_against_the_oracles_golden():
    """Configuration 2 at full size: EVERY pixel of the 1920x1080 frame of a 20 000-triangle mesh
    with materials against the hit ids the CPU oracle computed for it (a committed fixture: the
    oracle needs five minutes on eight cores for this frame, tests/golden/make_fullframe_golden.py),
    and the 8-bit colours of every 8th row within 1/255."""
    g = _golden("config2_sphere20k_1080p_f3")
    mesh = scenes.with_materials(scenes.sphere(20000), num_mats=4, seed=11)
    assert mesh.num_tris == int(g["triangles"])
    W, H = 1920, 1080
    fr = kernel.driver_frame(3)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF, want=("hit", "rgb8"))
    hit = out["hit"].cpu().numpy()
    assert (g["hit"] >= 0).sum() > 0.5 * W * H
    mism = int((hit != g["hit"]).sum())
    assert mism == 0, f"{mism} of {W * H} hit ids differ from the oracle's"
    rows = g["rgb8_rows"]
    d = np.abs(out["rgb8"].cpu().numpy()[rows].astype(int) - g["rgb8"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    sc.close()


def test_config3_whole_4k_frame_against_the_oracles_golden():
    """Configuration 3 at full size: every one of the 8 294 400 primary rays of the 69 451-triangle
    mesh against the oracle's hit ids (committed fixture; half an hour of CPU time)."""
    g = _golden("config3_torus69451_4k_f7")
    mesh = scenes.torus(69451)
    assert mesh.num_tris == int(g["triangles"])
    W, H = 3840, 2160
    fr = kernel.driver_frame(7)
    sc = kernel.Scene(mesh)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    hit, _ = sc.trace_primary(W, H, TAN_HALF)
    mism = int((hit.cpu().numpy() != g["hit"]).sum())
    assert (g["hit"] >= 0).sum() > 0.2 * W * H
    assert mism == 0, f"{mism} of {W * H} hit ids differ from the oracle's"
    sc.close()


def test_config5_rotation_sequence_frames():
    """Config 5: teapot-class mesh, per-frame M3d transform over an orbit; a few frames of the
    360-frame sequence at 4K, sampled rows against the oracle, and no drift: frame k rendered
    after other frames equals frame k rendered first (the object-space master is never
    overwritten, unlike driver.c:110,202)."""
    mesh = scenes.torus(6320)
    W, H = 3840, 2160
    sc = kernel.Scene(mesh)
    first = {}
    for k in (0, 90, 181, 359, 0, 90):
        fr = kernel.driver_frame(k, 360)
        sc.transform(kernel.frame_matrix(fr, "obmat"))
        hit, _ = sc.trace_primary(W, H, TAN_HALF)
        if k in first:
            assert torch.equal(first[k], hit)
        else:
            first[k] = hit
            X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
            ref = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF,
                                y0=300, y1=H, ystep=450, shade=False)
            rows = np.arange(300, H, 450)
            assert np.array_equal(hit.cpu().numpy()[rows], ref["hit"][rows])
    sc.close()
