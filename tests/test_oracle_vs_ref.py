"""Pins the oracle's restated matrix/determinant routines, and the product's host-side ones,
bit for bit against the reference's own M3d_matrix_tools.c compiled unmodified
(oracle/_ref/libm3d_ref.so, built by oracle/Makefile where the reference checkout exists)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import _lib

R = O.ref_lib()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref/libm3d_ref.so not built")


def _m():
    return np.zeros((4, 4))


def test_det_3x3_bit_exact():
    rng = np.random.default_rng(1)
    L = O.lib()
    for scale in (1.0, 1e-6, 1e6):
        for _ in range(2000):
            A, B, Cc = (rng.normal(size=3) * scale for _ in range(3))
            assert R.M3d_det_3x3(O.d(A), O.d(B), O.d(Cc)) == L.orc_det_3x3(O.d(A), O.d(B), O.d(Cc))
    # singular and zero inputs
    z = np.zeros(3)
    a = np.array([1.0, 2.0, 3.0])
    assert L.orc_det_3x3(O.d(a), O.d(a), O.d(z)) == R.M3d_det_3x3(O.d(a), O.d(a), O.d(z)) == 0.0


def test_mat_mult_points_bit_exact_and_in_place():
    rng = np.random.default_rng(2)
    L = O.lib()
    for n in (1, 2, 17, 1000):
        m = rng.normal(size=(4, 4))
        x, y, z = rng.normal(size=n) * 30, rng.normal(size=n) * 30, rng.normal(size=n) * 30
        X1, Y1, Z1 = np.empty(n), np.empty(n), np.empty(n)
        R.M3d_mat_mult_points(O.d(X1), O.d(Y1), O.d(Z1), m.ctypes.data, O.d(x), O.d(y), O.d(z), n)
        x2, y2, z2 = x.copy(), y.copy(), z.copy()
        L.orc_mat_mult_points(O.d(x2), O.d(y2), O.d(z2), m.ctypes.data, O.d(x2), O.d(y2), O.d(z2), n)
        assert np.array_equal(X1, x2) and np.array_equal(Y1, y2) and np.array_equal(Z1, z2)


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_view_and_frame_matrices_bit_exact(which):
    rng = np.random.default_rng(3)
    P = _lib.load()
    cases = [(np.array([15.0, 10.0, 0.0]), np.zeros(3), np.array([15.0, 11.0, 0.0]))]
    for k in range(60):                        # the driver's orbit, driver.c:34-54
        a = 2 * np.pi * k / 60
        eye = np.array([15.0 * np.cos(a), 10.0, 15.0 * np.sin(a)])
        cases.append((eye, np.zeros(3), eye + np.array([0.0, 1.0, 0.0])))
    for _ in range(200):                       # both branches of M3d_view (|b| < p and not)
        eye = rng.normal(size=3) * 10
        coi = rng.normal(size=3) * 3
        up = eye + rng.normal(size=3)
        cases.append((eye, coi, up))
    cases.append((np.array([0.0, 20.0, 0.0]), np.zeros(3), np.array([1.0, 20.0, 0.0])))   # straight down
    for eye, coi, up in cases:
        v1, vi1, v2, vi2 = _m(), _m(), _m(), _m()
        s1 = R.M3d_view(v1.ctypes.data, vi1.ctypes.data, O.d(eye), O.d(coi), O.d(up))
        if which == "oracle":
            s2 = O.lib().orc_view(v2.ctypes.data, vi2.ctypes.data, O.d(eye), O.d(coi), O.d(up))
            ok2 = s2 == 1
        else:
            ok2 = P.rt_view(v2.ctypes.data, vi2.ctypes.data, O.d(eye), O.d(coi), O.d(up)) == 0
        assert (s1 == 1) == ok2
        if s1 == 1:
            assert np.array_equal(v1, v2) and np.array_equal(vi1, vi2)


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_movement_sequence_bit_exact(which):
    rng = np.random.default_rng(4)
    P = _lib.load()
    for _ in range(200):
        n = int(rng.integers(0, 7))
        types = rng.integers(0, 12, size=max(n, 1)).astype(np.int32)
        params = rng.uniform(0.2, 300, size=max(n, 1))
        a1, b1, a2, b2 = _m(), _m(), _m(), _m()
        R.M3d_make_movement_sequence_matrix(a1.ctypes.data, b1.ctypes.data, n, O.i(types), O.d(params))
        if which == "oracle":
            assert O.lib().orc_movement_sequence(a2.ctypes.data, b2.ctypes.data, n, O.i(types), O.d(params)) == 1
        else:
            assert P.rt_movement_sequence(a2.ctypes.data, b2.ctypes.data, n, O.i(types), O.d(params)) == 0
        assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    bad = np.array([12], np.int32)
    one = np.array([1.0])
    assert O.lib().orc_movement_sequence(_m().ctypes.data, _m().ctypes.data, 1, O.i(bad), O.d(one)) == 0
    assert P.rt_movement_sequence(_m().ctypes.data, _m().ctypes.data, 1, O.i(bad), O.d(one)) != 0


def test_driver_frame_setup_matches_reference_composition():
    """driver.c:34-107 recomposed from the reference's own routines equals both restatements."""
    from raytracer_amd import kernel
    for frame in (0, 1, 7, 29, 59):
        a = 2 * np.pi * frame / 60
        eye = np.array([15.0 * np.cos(a), 10.0, 15.0 * np.sin(a)])
        coi = np.zeros(3)
        up = np.array([eye[0], eye[1] + 1, eye[2]])
        vm, vi, m, mi, obmat, obinv = (_m() for _ in range(6))
        R.M3d_view(vm.ctypes.data, vi.ctypes.data, O.d(eye), O.d(coi), O.d(up))
        t = np.array([3], np.int32)
        p = np.array([270.0])
        R.M3d_make_movement_sequence_matrix(m.ctypes.data, mi.ctypes.data, 1, O.i(t), O.d(p))
        R.M3d_mat_mult(obmat.ctypes.data, vm.ctypes.data, m.ctypes.data)
        R.M3d_mat_mult(obinv.ctypes.data, mi.ctypes.data, vi.ctypes.data)
        l0, l1 = np.zeros(3), np.zeros(3)
        R.M3d_mat_mult_pt(O.d(l0), vm.ctypes.data, O.d(np.array([0.0, 15.0 + frame, 0.0])))
        R.M3d_mat_mult_pt(O.d(l1), vm.ctypes.data, O.d(np.array([-10.0, 10.0, 0.0])))
        of = O.frame_setup(frame, 60)
        pf = kernel.driver_frame(frame, 60)
        for f in (of, pf):
            assert np.array_equal(O.arr4(f.obmat), obmat) and np.array_equal(O.arr4(f.obinv), obinv)
            assert np.array_equal(np.array(f.light_in_eye_space[0][:]), l0)
            assert np.array_equal(np.array(f.light_in_eye_space[1][:]), l1)
            assert f.num_lights == 2
        assert bytes(of) == bytes(pf)


def test_cross_dot_norm_twins():
    """The oracle's face-normal / Phong helpers use M3d_x_product, M3d_dot_product, M3d_norm
    formulas; check the same expressions against the compiled reference."""
    rng = np.random.default_rng(5)
    for _ in range(500):
        a, b = rng.normal(size=3), rng.normal(size=3)
        r = np.zeros(3)
        R.M3d_x_product(O.d(r), O.d(a), O.d(b))
        mine = np.array([a[1] * b[2] - b[1] * a[2], b[0] * a[2] - a[0] * b[2], a[0] * b[1] - b[0] * a[1]])
        assert np.array_equal(r, mine)
        assert R.M3d_dot_product(O.d(a), O.d(b)) == a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
        n = np.zeros(3)
        R.M3d_norm(O.d(n), O.d(a))
        mag = np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])
        assert np.array_equal(n, a / mag)


def test_oracle_dot_and_norm_are_the_references():
    """The oracle's own C helpers behind Light_Model (dot3, norm3 in rt_oracle.c), not a numpy
    restatement of them, against the compiled M3d_dot_product / M3d_norm: bit for bit, zero
    vector included."""
    rng = np.random.default_rng(17)
    L = O.lib()
    vecs = [rng.normal(size=3) * 10.0 ** rng.integers(-8, 8) for _ in range(2000)] + [np.zeros(3), np.array([0.0, -0.0, 0.0])]
    for k, a in enumerate(vecs):
        b = vecs[(k * 7 + 3) % len(vecs)]
        assert L.orc_dot3(O.d(a), O.d(b)) == R.M3d_dot_product(O.d(a), O.d(b))
        mine, ref = np.full(3, 7.0), np.full(3, 7.0)
        assert L.orc_norm3(O.d(mine), O.d(a)) == R.M3d_norm(O.d(ref), O.d(a))
        assert np.array_equal(mine, ref)          # untouched (7, 7, 7) when the vector is zero


def test_face_normals_of_both_loaders_are_m3d_x_product(tmp_path):
    """Faces without vn get one un-normalised normal AB x AC from the raw coordinates
    (obj_reader.c:51-82 through M3d_x_product): the product's loader (rt_obj_load) and the
    oracle's (obj_oracle.load) against the compiled reference's M3d_x_product."""
    import ctypes as C
    from oracle import obj_oracle
    from raytracer_amd import _lib
    rng = np.random.default_rng(23)
    nv, nf = 60, 40
    P = rng.normal(size=(nv, 3)) * 3.0
    faces = [rng.choice(nv, size=3, replace=False) + 1 for _ in range(nf)]
    path = tmp_path / "soup.obj"
    # repr() round-trips a double exactly through the loaders' decimal parsing
    path.write_text("".join(f"v {repr(float(p[0]))} {repr(float(p[1]))} {repr(float(p[2]))}\n" for p in P) +
                    "".join(f"f {f[0]} {f[1]} {f[2]}\n" for f in faces))
    want = []
    for f in faces:
        A, B, Cc = (P[i - 1] for i in f)
        AB, AC, r = np.ascontiguousarray(B - A), np.ascontiguousarray(Cc - A), np.zeros(3)
        R.M3d_x_product(O.d(r), O.d(AB), O.d(AC))
        want.append(r)
    want = np.array(want)
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rt_obj_load(str(path).encode(), C.byref(h)) == 0
    v = _lib.RtObjView()
    assert lib.rt_obj_get(h, C.byref(v)) == 0
    assert v.num_vn == nf and v.num_tris == nf
    got = np.stack([np.ctypeslib.as_array(getattr(v, k), shape=(nf + 1,))[1:] for k in ("xn", "yn", "zn")], 1).copy()
    An = np.ctypeslib.as_array(v.An, shape=(nf,)).copy()
    lib.rt_obj_free(h)
    assert np.array_equal(An, np.arange(1, nf + 1))
    assert np.array_equal(got, want)
    om = obj_oracle.load(str(path))
    assert np.array_equal(np.stack([om["xn"][1:], om["yn"][1:], om["zn"][1:]], 1), want)
