"""ctypes wrapper of the CPU oracle (oracle/rt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py import this
module; nothing under raytracer_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "librt_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libm3d_ref.so")
MAX_LIGHTS = 100

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)
Mat4 = (C.c_double * 4) * 4


class OrcTexture(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("rgb", C.c_void_p)]


class OrcMaterial(C.Structure):
    _fields_ = [("Ka", C.c_double * 3), ("Kd", C.c_double * 3), ("Ks", C.c_double * 3),
                ("map_Ka", C.c_int), ("map_Kd", C.c_int), ("map_Ks", C.c_int)]


class OrcScene(C.Structure):
    _fields_ = ([("num_tris", C.c_int)]
                + [(n, ip) for n in ("A", "B", "C", "An", "Bn", "Cn", "At", "Bt", "Ct", "mtl")]
                + [(n, dp) for n in ("x", "y", "z", "xn", "yn", "zn", "u", "v")]
                + [("num_mats", C.c_int), ("mats", C.POINTER(OrcMaterial)),
                   ("num_textures", C.c_int), ("textures", C.POINTER(OrcTexture))])


class OrcFrame(C.Structure):
    _fields_ = [("num_lights", C.c_int),
                ("light_in_eye_space", (C.c_double * 3) * MAX_LIGHTS),
                ("light_rgb", (C.c_double * 3) * MAX_LIGHTS),
                ("light_attenuation", (C.c_double * 2) * MAX_LIGHTS),
                ("light_max_dist", C.c_double * MAX_LIGHTS),
                ("obmat", Mat4), ("obinv", Mat4)]


def build(force: bool = False) -> None:
    """(Re)builds librt_oracle.so and, when the reference checkout exists, oracle/_ref."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(
            os.path.join(_HERE, "rt_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "librt_oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(REF_LIB)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.orc_det_3x3.restype = C.c_double
        L.orc_det_3x3.argtypes = [dp, dp, dp]
        L.orc_intersect_single_triangle.restype = C.c_double
        L.orc_intersect_single_triangle.argtypes = [dp] * 6
        L.orc_intersect_all_triangles.restype = C.c_int
        L.orc_intersect_all_triangles.argtypes = [dp, dp, dp, dp, ip, ip, ip, dp, dp, dp, C.c_int]
        L.orc_scene_intersect.restype = C.c_int
        L.orc_scene_intersect.argtypes = [C.POINTER(OrcScene), dp, dp, dp, dp, dp]
        L.orc_mat_mult_points.restype = None
        L.orc_mat_mult_points.argtypes = [dp, dp, dp, C.c_void_p, dp, dp, dp, C.c_int]
        L.orc_mat_mult.restype = None
        L.orc_mat_mult.argtypes = [C.c_void_p] * 3
        L.orc_mat_mult_pt.restype = None
        L.orc_mat_mult_pt.argtypes = [dp, C.c_void_p, dp]
        L.orc_movement_sequence.restype = C.c_int
        L.orc_movement_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_int, ip, dp]
        L.orc_view.restype = C.c_int
        L.orc_view.argtypes = [C.c_void_p, C.c_void_p, dp, dp, dp]
        L.orc_driver_frame_setup.restype = None
        L.orc_driver_frame_setup.argtypes = [C.c_int, C.c_int, C.POINTER(OrcFrame)]
        L.orc_center_and_scale.restype = None
        L.orc_center_and_scale.argtypes = [dp, dp, dp, C.c_int, C.c_void_p]
        L.orc_light_model.restype = None
        L.orc_light_model.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcFrame)] + [dp] * 7 + [C.c_void_p]
        L.orc_interpolate_normal.restype = None
        L.orc_interpolate_normal.argtypes = [C.POINTER(OrcScene), C.c_int, dp, C.c_void_p, dp]
        L.orc_quantize.restype = C.c_uint8
        L.orc_quantize.argtypes = [C.c_double]
        L.orc_render_rows.restype = C.c_long
        L.orc_render_rows.argtypes = [C.POINTER(OrcScene), C.POINTER(OrcFrame), C.c_int, C.c_int,
                                      C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_place_pixels.restype = None
        L.orc_place_pixels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_num_threads.restype = C.c_int
        L.orc_dot3.restype = C.c_double
        L.orc_dot3.argtypes = [dp, dp]
        L.orc_norm3.restype = C.c_int
        L.orc_norm3.argtypes = [dp, dp]
        L.orc_trace_rays.restype = None
        L.orc_trace_rays.argtypes = [C.POINTER(OrcScene), C.c_long, dp, dp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def ref_lib():
    """The reference's own M3d_matrix_tools.c, compiled unmodified (None when not built)."""
    if not os.path.exists(REF_LIB):
        try:
            build()
        except Exception:
            return None
    if not os.path.exists(REF_LIB):
        return None
    R = C.CDLL(REF_LIB)
    R.M3d_det_3x3.restype = C.c_double
    R.M3d_det_3x3.argtypes = [dp, dp, dp]
    R.M3d_mat_mult_points.argtypes = [dp, dp, dp, C.c_void_p, dp, dp, dp, C.c_int]
    R.M3d_mat_mult.argtypes = [C.c_void_p] * 3
    R.M3d_mat_mult_pt.argtypes = [dp, C.c_void_p, dp]
    R.M3d_view.argtypes = [C.c_void_p, C.c_void_p, dp, dp, dp]
    R.M3d_make_movement_sequence_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_int, ip, dp]
    R.M3d_x_product.argtypes = [dp, dp, dp]
    R.M3d_norm.argtypes = [dp, dp]
    R.M3d_dot_product.restype = C.c_double
    R.M3d_dot_product.argtypes = [dp, dp]
    return R


def d(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(dp)


def i(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ip)


class Scene:
    """Keeps the numpy arrays alive behind an OrcScene.  `mesh` is any object with the
    attributes of raytracer_amd.scenes.Mesh; x/y/z may be overridden (eye-space copies)."""

    def __init__(self, mesh, x=None, y=None, z=None):
        self.keep = []
        s = OrcScene()
        s.num_tris = mesh.num_tris
        for name in ("A", "B", "C", "An", "Bn", "Cn", "At", "Bt", "Ct", "mtl"):
            a = getattr(mesh, name)
            if a is not None:
                a = np.ascontiguousarray(a, np.int32)
                self.keep.append(a)
            setattr(s, name, i(a))
        xs = {"x": x, "y": y, "z": z}
        for name in ("x", "y", "z", "xn", "yn", "zn", "u", "v"):
            a = xs.get(name)
            if a is None:
                a = getattr(mesh, name)
            a = np.ascontiguousarray(a, np.float64)
            self.keep.append(a)
            setattr(s, name, d(a))
        nm = len(mesh.Ka)
        mats = (OrcMaterial * max(nm, 1))()
        for k in range(nm):
            for c in range(3):
                mats[k].Ka[c] = mesh.Ka[k][c]
                mats[k].Kd[c] = mesh.Kd[k][c]
                mats[k].Ks[c] = mesh.Ks[k][c]
            mats[k].map_Ka = -1 if mesh.map_Ka is None else int(mesh.map_Ka[k])
            mats[k].map_Kd = -1 if mesh.map_Kd is None else int(mesh.map_Kd[k])
            mats[k].map_Ks = -1 if mesh.map_Ks is None else int(mesh.map_Ks[k])
        s.num_mats = nm
        s.mats = mats
        nt = len(mesh.textures)
        tex = (OrcTexture * max(nt, 1))()
        for k, t in enumerate(mesh.textures):
            t = np.ascontiguousarray(t, np.uint8)
            self.keep.append(t)
            tex[k].w, tex[k].h, tex[k].rgb = t.shape[1], t.shape[0], t.ctypes.data
        s.num_textures = nt
        s.textures = tex
        self.keep += [mats, tex]
        self.s = s


def mat4(m) -> np.ndarray:
    return np.ascontiguousarray(m, dtype=np.float64).reshape(4, 4)


def frame_setup(frame_number: int, frames_per_turn: int = 60) -> OrcFrame:
    f = OrcFrame()
    lib().orc_driver_frame_setup(frame_number, frames_per_turn, C.byref(f))
    return f


def frame_from(src) -> OrcFrame:
    """Copies any structure with the rt_frame / OrcFrame fields into an OrcFrame."""
    f = OrcFrame()
    C.memmove(C.byref(f), C.byref(src), C.sizeof(OrcFrame))
    return f


def arr4(cm) -> np.ndarray:
    return np.array([[cm[r][c] for c in range(4)] for r in range(4)])


def transform(mesh, m):
    """orc_mat_mult_points over all num_v+1 slots; returns new (x, y, z)."""
    m = mat4(m)
    n = mesh.num_v + 1
    X, Y, Z = np.empty(n), np.empty(n), np.empty(n)
    lib().orc_mat_mult_points(d(X), d(Y), d(Z), m.ctypes.data, d(mesh.x), d(mesh.y), d(mesh.z), n)
    return X, Y, Z


def render_rows(scene: Scene, frame: OrcFrame, W: int, H: int, tan_half: float, y0: int = 0,
                y1: int | None = None, ystep: int = 1, shade: bool = True, nthreads: int = 0):
    """Full-image-sized outputs (rows not visited keep hit = -2)."""
    y1 = H if y1 is None else y1
    hit = np.full((H, W), -2, dtype=np.int32)
    tuv = np.zeros((H, W, 3))
    argb = np.zeros((H, W, 3))
    rgb8 = np.zeros((H, W, 3), dtype=np.uint8)
    shadows = lib().orc_render_rows(C.byref(scene.s), C.byref(frame), W, H, float(tan_half), y0, y1,
                                    ystep, 1 if shade else 0, nthreads, hit.ctypes.data,
                                    tuv.ctypes.data, argb.ctypes.data, rgb8.ctypes.data)
    return {"hit": hit, "tuv": tuv, "argb": argb, "rgb8": rgb8, "shadow_rays": int(shadows)}


def intersect(scene: Scene, S, E):
    S = np.ascontiguousarray(S, np.float64)
    E = np.ascontiguousarray(E, np.float64)
    uv, point, t = np.zeros(2), np.zeros(3), C.c_double(0)
    idx = lib().orc_scene_intersect(C.byref(scene.s), d(S), d(E), d(uv), d(point), C.byref(t))
    return idx, t.value, uv, point


def trace_rays(scene: Scene, S, E, nthreads: int = 0):
    S = np.ascontiguousarray(S, np.float64)
    E = np.ascontiguousarray(E, np.float64)
    n = S.shape[0]
    hit = np.full(n, -1, np.int32)
    tuv = np.zeros((n, 3))
    point = np.zeros((n, 3))
    lib().orc_trace_rays(C.byref(scene.s), n, d(S), d(E), hit.ctypes.data, tuv.ctypes.data,
                         point.ctypes.data, nthreads)
    return hit, tuv, point


def place_pixels(rgb8: np.ndarray) -> np.ndarray:
    H, W, _ = rgb8.shape
    src = np.ascontiguousarray(rgb8, np.uint8)
    out = np.empty_like(src)
    lib().orc_place_pixels(src.ctypes.data, W, H, out.ctypes.data)
    return out
