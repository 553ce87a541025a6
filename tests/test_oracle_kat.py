"""First-principles known-answer tests of the oracle's intersection, reduction, shading and
pixel rules.  The reference ships no tests or golden vectors for these parts, so they are
checked against answers derived by hand from its formulas (file:line in each test)."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import scenes
from conftest import TAN_HALF


def _one_tri(A, B, Cc, extra=()):
    pts = [np.zeros(3), np.asarray(A, float), np.asarray(B, float), np.asarray(Cc, float)]
    tris = [(1, 2, 3)]
    for t in extra:
        base = len(pts)
        pts += [np.asarray(p, float) for p in t]
        tris.append((base, base + 1, base + 2))
    P = np.array(pts)
    T = np.array(tris, np.int32)
    n = len(T)
    nrm = np.zeros((n + 1, 3))
    nrm[1:] = [0, 0, -1]
    idn = np.arange(1, n + 1, dtype=np.int32)
    return scenes.Mesh(x=P[:, 0].copy(), y=P[:, 1].copy(), z=P[:, 2].copy(),
                       xn=nrm[:, 0].copy(), yn=nrm[:, 1].copy(), zn=nrm[:, 2].copy(),
                       A=T[:, 0].copy(), B=T[:, 1].copy(), C=T[:, 2].copy(),
                       An=idn, Bn=idn.copy(), Cn=idn.copy())


def test_centroid_ray_gives_one_third():
    # triangle_intersection.cu:78-86: u pairs with B, v with C
    A, B, Cc = (0, 0, 5), (3, 0, 5), (0, 3, 5)
    sc = O.Scene(_one_tri(A, B, Cc))
    cen = (np.array(A) + B + Cc) / 3.0
    idx, t, uv, p = O.intersect(sc, np.zeros(3), cen * 0.5)     # E halfway: t = 2
    assert idx == 0 and t == pytest.approx(2.0, abs=1e-15)
    assert uv[0] == pytest.approx(1 / 3, abs=1e-15) and uv[1] == pytest.approx(1 / 3, abs=1e-15)
    assert np.allclose(p, cen, atol=1e-14)                       # :185-187


def test_u_goes_with_B_and_v_with_C():
    A, B, Cc = (0, 0, 5), (4, 0, 5), (0, 4, 5)
    sc = O.Scene(_one_tri(A, B, Cc))
    idx, t, uv, _ = O.intersect(sc, np.zeros(3), np.array([3.0, 0.5, 5.0]))
    assert idx == 0 and uv[0] == pytest.approx(0.75) and uv[1] == pytest.approx(0.125)


def test_inclusive_edges_and_vertices():
    # :81,87,91 reject only u<0, u>1, v<0, v>1, u+v>1: boundary points are hits
    A, B, Cc = (0, 0, 2), (2, 0, 2), (0, 2, 2)
    sc = O.Scene(_one_tri(A, B, Cc))
    for target in [(0, 0, 2), (2, 0, 2), (0, 2, 2), (1, 0, 2), (0, 1, 2), (1, 1, 2)]:
        idx, _, _, _ = O.intersect(sc, np.zeros(3), np.array(target, float))
        assert idx == 0, target
    for target in [(-1e-9, 1, 2), (1, -1e-9, 2), (1 + 1e-9, 1, 2)]:
        idx, _, _, _ = O.intersect(sc, np.zeros(3), np.array(target, float))
        assert idx == -1, target


def test_parallel_ray_and_triangle_behind_origin_miss():
    A, B, Cc = (0, 0, 5), (3, 0, 5), (0, 3, 5)
    sc = O.Scene(_one_tri(A, B, Cc))
    assert O.intersect(sc, np.zeros(3), np.array([1.0, 1.0, 0.0]))[0] == -1      # den == 0, :67
    assert O.intersect(sc, np.zeros(3), np.array([-0.5, -0.5, -5.0]))[0] == -1   # t < 0, :74
    # t == 0 exactly (origin on the triangle) is rejected by the host loop's dist > 0, :169
    assert O.intersect(sc, np.array([1.0, 1.0, 5.0]), np.array([1.0, 1.0, 9.0]))[0] == -1


def test_no_near_clip_and_no_far_clip():
    # hits with 0 < t < 1 (in front of the image plane) and t >> 1 both count (driver.c:121-135)
    near = _one_tri((-1, -1, 0.5), (3, -1, 0.5), (-1, 3, 0.5))
    assert O.intersect(O.Scene(near), np.zeros(3), np.array([0.0, 0.0, 100.0]))[1] == pytest.approx(0.005)
    far = _one_tri((-1e6, -1e6, 1e6), (3e6, -1e6, 1e6), (-1e6, 3e6, 1e6))
    assert O.intersect(O.Scene(far), np.zeros(3), np.array([0.0, 0.0, 1.0]))[1] == pytest.approx(1e6)


def test_tie_goes_to_lowest_index():
    # two coincident triangles: strict '<' at :169 keeps the first
    tri = ((0, 0, 5), (3, 0, 5), (0, 3, 5))
    sc = O.Scene(_one_tri(*tri, extra=[tri, tri]))
    assert O.intersect(sc, np.zeros(3), np.array([0.5, 0.5, 2.5]))[0] == 0
    # shared edge between two coplanar triangles: both contain the edge point, lower index wins
    m = _one_tri((0, 0, 5), (2, 0, 5), (0, 2, 5), extra=[((2, 0, 5), (2, 2, 5), (0, 2, 5))])
    assert O.intersect(O.Scene(m), np.zeros(3), np.array([1.0, 1.0, 5.0]))[0] == 0


def test_closest_wins_regardless_of_order():
    near = ((-1, -1, 3), (3, -1, 3), (-1, 3, 3))
    far = ((-1, -1, 6), (3, -1, 6), (-1, 3, 6))
    for order in ((near, far), (far, near)):
        sc = O.Scene(_one_tri(*order[0], extra=[order[1]]))
        idx, t, _, _ = O.intersect(sc, np.zeros(3), np.array([0.0, 0.0, 1.0]))
        assert t == pytest.approx(3.0) and idx == (0 if order[0] is near else 1)


def test_miss_leaves_uv_and_point_untouched():
    import ctypes as C
    A, B, Cc = (0, 0, 5), (3, 0, 5), (0, 3, 5)
    m = _one_tri(A, B, Cc)
    uv, point = np.full(2, 9.0), np.full(3, 9.0)
    S, E = np.zeros(3), np.array([-1.0, -1.0, 1.0])
    r = O.lib().orc_intersect_all_triangles(O.d(S), O.d(E), O.d(uv), O.d(point), O.i(m.A), O.i(m.B),
                                            O.i(m.C), O.d(m.x), O.d(m.y), O.d(m.z), 1)
    assert r == -1 and np.all(uv == 9.0) and np.all(point == 9.0)


def test_quantize_truncates():
    # graphic_tools.c:106-113: c*255 then truncation; miss colour 0.5 -> 127 (driver.c:136-139)
    q = O.lib().orc_quantize
    assert q(0.5) == 127 and q(1.0) == 255 and q(0.0) == 0 and q(0.999) == 254 and q(0.2) == 51


def test_pixel_placement_row0_black_and_ypix0_dropped():
    # driver.c:31-32,186-190
    H, W = 5, 3
    rgb = np.arange(H * W * 3, dtype=np.uint8).reshape(H, W, 3) + 1
    scr = O.place_pixels(rgb)
    assert np.all(scr[0] == 0)
    for y in range(1, H):
        assert np.array_equal(scr[H - y], rgb[y])


def _frame(lights):
    f = O.OrcFrame()
    f.num_lights = len(lights)
    for i, (pos, rgb, att, mx) in enumerate(lights):
        for k in range(3):
            f.light_in_eye_space[i][k] = pos[k]
            f.light_rgb[i][k] = rgb[k]
        f.light_attenuation[i][0], f.light_attenuation[i][1] = att
        f.light_max_dist[i] = mx
    for r in range(4):
        f.obinv[r][r] = 1.0
        f.obmat[r][r] = 1.0
    return f


def _shade(mesh, frame, p, n, Ka=(1, 1, 1), Kd=(1, 1, 1), Ks=(1, 1, 1)):
    sc = O.Scene(mesh)
    out = np.zeros(3)
    args = [np.array(v, float) for v in (Ka, Kd, Ks, (0, 0, 0), p, n)]
    O.lib().orc_light_model(sc.s, frame, *[O.d(a) for a in args], O.d(out), None)
    return out


def test_light_model_hand_computed():
    # light_model.c:17-151 for one unshadowed light, by hand
    mesh = _one_tri((100, 100, 100), (101, 100, 100), (100, 101, 100))      # far away: no shadow
    p, n = (0.0, 0.0, 10.0), (0.0, 0.0, -2.0)
    light = (0.0, 3.0, 6.0)
    f = _frame([(light, (1.0, 0.5, 0.25), (0.045, 0.0075), 100.0)])
    got = _shade(mesh, f, p, n, Ka=(0.3, 0.6, 0.9), Kd=(0.5, 0.5, 0.5), Ks=(0.2, 0.4, 0.8))
    N = np.array([0, 0, -1.0])
    L = np.array(light) - p
    ln = math.sqrt(L @ L)
    L = L / ln
    ndl = N @ L
    att = 1 + 0.045 * ln + 0.0075 * ln * ln
    V = -np.array(p) / 10.0
    Rv = 2 * ndl * N - L
    fs = max(Rv @ V, 0.0) ** 50 if Rv @ V > 0 else 0.0
    exp = []
    for k, (ka, kd, ks, lr) in enumerate(zip((0.3, 0.6, 0.9), (0.5,) * 3, (0.2, 0.4, 0.8), (1.0, 0.5, 0.25))):
        diff = max(ndl * kd * 0.5 * lr, 0.0)
        spec = fs * ks * (1.0 - 0.5 - 0.2) * lr
        exp.append(min(0.2 * ka + (diff + spec) / att, 1.0))
    assert np.allclose(got, exp, rtol=0, atol=1e-15)


def test_light_beyond_max_dist_and_clamp_and_ambient_only():
    mesh = _one_tri((100, 100, 100), (101, 100, 100), (100, 101, 100))
    p, n = (0.0, 0.0, 10.0), (0.0, 0.0, -1.0)
    f = _frame([((0.0, 0.0, -50.0), (1, 1, 1), (0.0, 0.0), 50.0)])           # len 60 > 50
    assert np.allclose(_shade(mesh, f, p, n), [0.2, 0.2, 0.2])
    f = _frame([((0.0, 0.0, 9.0), (5, 5, 5), (0.0, 0.0), 100.0)])            # very bright: clamp :147
    assert np.array_equal(_shade(mesh, f, p, n), [1.0, 1.0, 1.0])
    f = _frame([])
    assert np.allclose(_shade(mesh, f, p, n, Ka=(0.5, 1.0, 0.0)), [0.1, 0.2, 0.0])


def test_zero_normal_contributes_nothing():
    mesh = _one_tri((100, 100, 100), (101, 100, 100), (100, 101, 100))
    f = _frame([((0.0, 0.0, 5.0), (1, 1, 1), (0.0, 0.0), 100.0)])
    assert np.allclose(_shade(mesh, f, (0.0, 0.0, 10.0), (0.0, 0.0, 0.0)), [0.2, 0.2, 0.2])


def test_shadow_blocker_between_and_beyond_the_light():
    # light_model.c:60-76: any hit with t > 0 shadows -- also an occluder BEHIND the light
    p, n = (0.0, 0.0, 10.0), (0.0, 0.0, -1.0)
    light = (0.0, 0.0, 5.0)
    f = _frame([(light, (1, 1, 1), (0.0, 0.0), 100.0)])
    between = _one_tri((-1, -1, 7), (3, -1, 7), (-1, 3, 7))
    beyond = _one_tri((-1, -1, 2), (3, -1, 2), (-1, 3, 2))
    behind_p = _one_tri((-1, -1, 12), (3, -1, 12), (-1, 3, 12))
    assert np.allclose(_shade(between, f, p, n), [0.2] * 3)
    assert np.allclose(_shade(beyond, f, p, n), [0.2] * 3)
    lit = _shade(behind_p, f, p, n)
    assert np.all(lit > 0.2)
    # the surface the point lies on does not shadow itself: origin is 0.001 along L (:61-63)
    own = _one_tri((-1, -1, 10), (3, -1, 10), (-1, 3, 10))
    assert np.allclose(_shade(own, f, p, n), lit)


def test_normal_interpolation_uses_inverse_transpose():
    # light_model.c:178-185: normal_j = sum_i Ntemp_i * obinv[i][j]
    m = _one_tri((0, 0, 5), (1, 0, 5), (0, 1, 5))
    m.xn = np.array([0.0, 1.0, 0.0, 0.0]); m.yn = np.array([0.0, 0.0, 1.0, 0.0]); m.zn = np.array([0.0, 0.0, 0.0, 1.0])
    m.An, m.Bn, m.Cn = (np.array([k], np.int32) for k in (1, 2, 3))
    sc = O.Scene(m)
    obinv = np.arange(16, dtype=float).reshape(4, 4) + 1
    uv = np.array([0.25, 0.5])
    out = np.zeros(3)
    O.lib().orc_interpolate_normal(sc.s, 0, O.d(uv), obinv.ctypes.data, O.d(out))
    nt = np.array([0.25, 0.25, 0.5])            # (1-u-v)*An + u*Bn + v*Cn
    assert np.allclose(out, nt @ obinv[:3, :3])


def test_ray_generation_integer_halves():
    # driver.c:131-133: W/2 and H/2 are integer divisions
    W, H = 7, 5
    big = _one_tri((-100, -100, 10), (300, -100, 10), (-100, 300, 10))
    sc = O.Scene(big)
    f = _frame([])
    r = O.render_rows(sc, f, W, H, TAN_HALF, shade=False)
    ez = (W // 2) / TAN_HALF
    for yp in range(H):
        for xp in range(W):
            t = r["tuv"][yp, xp, 0]
            assert t == pytest.approx(10.0 / ez)
            # hit point = t*E: x = t*(xp - 3), y = t*(yp - 2)
    assert np.all(r["hit"] == 0)


def test_center_and_scale():
    # obj_reader.c:239-289
    m = scenes.triangle_soup(50, seed=3, normalize=False)
    x, y, z = m.x.copy(), m.y.copy(), m.z.copy()
    O.lib().orc_center_and_scale(O.d(x), O.d(y), O.d(z), m.num_v, None)
    ext = max(np.ptp(x[1:]), np.ptp(y[1:]), np.ptp(z[1:]))
    assert ext == pytest.approx(15.0, rel=1e-12)
    assert abs(x[1:].mean()) < 1e-12 and abs(y[1:].mean()) < 1e-12
    m2 = m.copy()
    scenes.center_and_scale(m2)
    assert np.array_equal(m2.x, x) and np.array_equal(m2.y, y) and np.array_equal(m2.z, z)
