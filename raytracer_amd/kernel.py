"""Host-side mirror of the reference's device interface for Python callers.

`mat_mult_device` and `intersect_all_triangles_device` keep the names, argument order and
in-place behaviour of kernel.h:10-17 (numpy arrays instead of raw pointers).  `Scene` wraps
the additive batched API of include/rt_amd.h over a device-resident mesh; PyTorch is used
only to own device memory for outputs.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import RtFrame, RtProfile, RtRows, RtSequenceStats, check
from .scenes import Mesh


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_lib.c_double_p)


def _ip(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_lib.c_int_p)


def _mat(m) -> np.ndarray:
    m = np.ascontiguousarray(m, dtype=np.float64)
    assert m.shape == (4, 4)
    return m


# ------------------------------------------------------------------ reference API

def mat_mult_device(X, Y, Z, m, x, y, z, numpoints: int) -> None:
    """kernel.h:10-14.  Outputs may be the input arrays (driver.c:110)."""
    lib = _lib.load()
    m = _mat(m)
    lib.mat_mult_device(_dp(X), _dp(Y), _dp(Z), m.ctypes.data, _dp(x), _dp(y), _dp(z), int(numpoints))


class TriangleArray:
    """A host array laid out like the reference's `Triangle tris[]` (obj_reader.h:25-39,
    488 bytes each), built from a Mesh so the kernel.h entry point can be exercised."""

    def __init__(self, mesh: Mesh, stride: int = 488):
        n = mesh.num_tris
        self.stride = stride
        self.buf = np.zeros(max(n, 1) * stride, dtype=np.uint8)
        ints = self.buf.view(np.int32).reshape(max(n, 1), stride // 4)
        ints[:n, 0], ints[:n, 1], ints[:n, 2] = mesh.A, mesh.B, mesh.C
        if mesh.At is not None:
            ints[:n, 3], ints[:n, 4], ints[:n, 5] = mesh.At, mesh.Bt, mesh.Ct
        ints[:n, 6], ints[:n, 7], ints[:n, 8] = mesh.An, mesh.Bn, mesh.Cn
        mtl = mesh.mtl if mesh.mtl is not None else np.full(n, -1, np.int32)
        ints[:n, 10] = mtl                       # Material.index at byte 40
        dbl = self.buf.view(np.float64).reshape(max(n, 1), stride // 8)
        for i in range(n):                       # Ka@304, Kd@328, Ks@352
            if mtl[i] >= 0:
                dbl[i, 38:41] = mesh.Ka[mtl[i]]
                dbl[i, 41:44] = mesh.Kd[mtl[i]]
                dbl[i, 44:47] = mesh.Ks[mtl[i]]
        self.num_tris = n

    @property
    def ptr(self):
        return self.buf.ctypes.data


def intersect_all_triangles_device(S, E, uv, point, tris: TriangleArray, x, y, z, num_tris: int) -> int:
    """kernel.h:16-17.  Returns the hit index or -1; uv and point are written on a hit only."""
    lib = _lib.load()
    S = np.ascontiguousarray(S, dtype=np.float64)
    E = np.ascontiguousarray(E, dtype=np.float64)
    return int(lib.intersect_all_triangles_device(_dp(S), _dp(E), _dp(uv), _dp(point), tris.ptr,
                                                  _dp(x), _dp(y), _dp(z), int(num_tris)))


# ------------------------------------------------------------------ frames

def driver_frame(frame_number: int, frames_per_turn: int = 60) -> RtFrame:
    """driver.c:34-107 for one frame (camera orbit, two lights, RX(270) object matrix)."""
    f = RtFrame()
    check(_lib.load().rt_driver_frame_setup(frame_number, frames_per_turn, C.byref(f)),
          "rt_driver_frame_setup")
    return f


def frame_matrix(f: RtFrame, which: str) -> np.ndarray:
    return np.array([[getattr(f, which)[r][c] for c in range(4)] for r in range(4)])


def band(H: int, nranks: int, rank: int) -> tuple[int, int]:
    y0, y1 = C.c_int(), C.c_int()
    check(_lib.load().rt_band(H, nranks, rank, C.byref(y0), C.byref(y1)), "rt_band")
    return y0.value, y1.value


def rows_part(H: int, parts: int = 1, part: int = 0, block: int = 16, y0: int = 0,
              y1: int | None = None) -> RtRows:
    """Row set `part` of `parts` of the rows [y0,y1), interleaved in blocks of `block` rows dealt
    forwards or backwards after the Thue-Morse sequence, the rows after the last complete round
    shared out evenly (rt_rows in include/rt_amd.h).  parts = 1 is the contiguous band."""
    return RtRows(y0, H if y1 is None else y1, block, parts, part)


def rows_count(r: RtRows) -> int:
    n = _lib.load().rt_rows_count(C.byref(r))
    check(min(n, 0), "rt_rows_count")
    return n


def rows_image_rows(r: RtRows) -> np.ndarray:
    """Image row of every local row of the set, ascending (the rule of rt_rows in
    include/rt_amd.h, restated in numpy)."""
    n = rows_count(r)
    total = r.y1 - r.y0
    full_rounds, tail = divmod(total, r.block * r.parts)

    def slot(rnd):
        odd_bits = np.array([bin(int(k)).count("1") & 1 for k in np.atleast_1d(rnd)], dtype=bool)   # Thue-Morse: F B B F B F F B
        return np.where(odd_bits, r.parts - 1 - r.part, r.part)

    lr = np.arange(n)
    rnd = lr // r.block
    whole = r.y0 + (rnd * r.parts + slot(rnd)) * r.block + lr % r.block
    run = r.y0 + full_rounds * r.parts * r.block + tail * int(slot(full_rounds)[0]) // r.parts + (lr - full_rounds * r.block)
    return np.where(rnd < full_rounds, whole, run)


# ------------------------------------------------------------------ batched API

class ObjFile:
    """A Wavefront .obj loaded by the library (rt_obj_load: obj_reader.c:628-726 semantics, quads
    split, centre + scale, page-locked SoA when a GPU is present).  `mesh()` is a host copy of
    its arrays for code that wants to look at them (the CPU oracle, triangle counts)."""

    def __init__(self, path: str):
        self.lib = _lib.load()
        self.path = path
        h = C.c_void_p()
        check(self.lib.rt_obj_load(os.fsencode(path), C.byref(h)), "rt_obj_load")
        self.h = h
        self.view = _lib.RtObjView()
        check(self.lib.rt_obj_get(self.h, C.byref(self.view)), "rt_obj_get")

    @property
    def num_tris(self) -> int:
        return int(self.view.num_tris)

    def mesh(self) -> Mesh:
        v = self.view

        def d(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.zeros(0)

        def i(ptr, n):
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(np.int32) if n else np.zeros(0, np.int32)

        nt, nm = v.num_tris, v.num_mats
        textures = []
        for k in range(self.lib.rt_obj_num_textures(self.h)):
            w, hh, px = C.c_int(), C.c_int(), C.c_void_p()
            check(self.lib.rt_obj_texture(self.h, k, C.byref(w), C.byref(hh), C.byref(px)), "rt_obj_texture")
            buf = (C.c_uint8 * (w.value * hh.value * 3)).from_address(px.value)
            textures.append(np.frombuffer(buf, np.uint8).reshape(hh.value, w.value, 3).copy())
        has_vt = v.num_vt > 0
        return Mesh(x=d(v.x, v.num_v + 1), y=d(v.y, v.num_v + 1), z=d(v.z, v.num_v + 1),
                    xn=d(v.xn, v.num_vn + 1), yn=d(v.yn, v.num_vn + 1), zn=d(v.zn, v.num_vn + 1),
                    u=d(v.u, v.num_vt + 1), v=d(v.v, v.num_vt + 1),
                    A=i(v.A, nt), B=i(v.B, nt), C=i(v.C, nt), An=i(v.An, nt), Bn=i(v.Bn, nt), Cn=i(v.Cn, nt),
                    At=i(v.At, nt) if has_vt else None, Bt=i(v.Bt, nt) if has_vt else None,
                    Ct=i(v.Ct, nt) if has_vt else None, mtl=i(v.mtl, nt),
                    Ka=d(v.Ka, 3 * nm).reshape(nm, 3), Kd=d(v.Kd, 3 * nm).reshape(nm, 3),
                    Ks=d(v.Ks, 3 * nm).reshape(nm, 3),
                    map_Ka=i(v.map_Ka, nm) if nm else None, map_Kd=i(v.map_Kd, nm) if nm else None,
                    map_Ks=i(v.map_Ks, nm) if nm else None, textures=textures,
                    name=f"{os.path.basename(self.path)} ({nt} triangles)")

    def close(self):
        if self.h:
            self.lib.rt_obj_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """Device-resident mesh + the batched trace/render calls (include/rt_amd.h, group 2)."""

    def __init__(self, mesh: Mesh | None = None, device: int = 0):
        import torch

        self.torch = torch
        self.lib = _lib.load()
        check(self.lib.rt_init(device), "rt_init")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        check(self.lib.rt_scene_create(C.byref(h)), "rt_scene_create")
        self.h = h
        self.mesh = None
        if isinstance(mesh, ObjFile):
            self.set_obj(mesh)
        elif mesh is not None:
            self.set_mesh(mesh)

    def set_obj(self, obj: "ObjFile") -> None:
        """The loaded file's arrays straight from the library's own (page-locked) buffers."""
        check(self.lib.rt_scene_set_obj(self.h, obj.h), "rt_scene_set_obj")

    @classmethod
    def adopt(cls, handle, device: int = 0) -> "Scene":
        """A Scene over an rt_scene handle that someone else owns (the scenes of a MultiScene)."""
        import torch

        self = cls.__new__(cls)
        self.torch = torch
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self.mesh = None
        self.owned = False
        return self

    def close(self):
        if self.h and getattr(self, "owned", True):
            self.lib.rt_scene_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mesh(self, mesh: Mesh) -> None:
        lib, h = self.lib, self.h
        self.mesh = mesh
        check(lib.rt_scene_set_vertices(h, mesh.num_v, _dp(mesh.x), _dp(mesh.y), _dp(mesh.z)), "set_vertices")
        check(lib.rt_scene_set_normals(h, mesh.num_vn, _dp(mesh.xn), _dp(mesh.yn), _dp(mesh.zn)), "set_normals")
        if mesh.num_vt > 0:
            check(lib.rt_scene_set_texcoords(h, mesh.num_vt, _dp(mesh.u), _dp(mesh.v)), "set_texcoords")
        if mesh.textures:
            n = len(mesh.textures)
            keep = [np.ascontiguousarray(t, dtype=np.uint8) for t in mesh.textures]
            w = np.array([t.shape[1] for t in keep], np.int32)
            hh = np.array([t.shape[0] for t in keep], np.int32)
            ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in keep])
            check(lib.rt_scene_set_textures(h, n, _ip(w), _ip(hh), ptrs), "set_textures")
        nm = len(mesh.Ka)
        if nm:
            Ka = np.ascontiguousarray(mesh.Ka, np.float64)
            Kd = np.ascontiguousarray(mesh.Kd, np.float64)
            Ks = np.ascontiguousarray(mesh.Ks, np.float64)
            maps = [None if a is None else np.ascontiguousarray(a, np.int32)
                    for a in (mesh.map_Ka, mesh.map_Kd, mesh.map_Ks)]
            check(lib.rt_scene_set_materials(h, nm, _dp(Ka), _dp(Kd), _dp(Ks), *[_ip(a) for a in maps]),
                  "set_materials")
        check(lib.rt_scene_set_triangles(h, mesh.num_tris, _ip(mesh.A), _ip(mesh.B), _ip(mesh.C),
                                         _ip(mesh.An), _ip(mesh.Bn), _ip(mesh.Cn),
                                         _ip(mesh.At), _ip(mesh.Bt), _ip(mesh.Ct), _ip(mesh.mtl)),
              "set_triangles")

    def transform(self, m) -> None:
        """mat_mult_device(x,y,z, m, x,y,z, num_v+1) on the device copy (driver.c:110)."""
        m = _mat(m)
        check(self.lib.rt_scene_transform(self.h, m.ctypes.data), "rt_scene_transform")

    def bake(self, m) -> None:
        m = _mat(m)
        check(self.lib.rt_scene_bake(self.h, m.ctypes.data), "rt_scene_bake")

    def eye_vertices(self):
        n = self.mesh.num_v + 1
        x, y, z = np.empty(n), np.empty(n), np.empty(n)
        check(self.lib.rt_scene_get_eye_vertices(self.h, _dp(x), _dp(y), _dp(z)), "get_eye_vertices")
        return x, y, z

    def _empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    def trace_primary(self, W: int, H: int, tan_half: float, y0: int = 0, y1: int | None = None):
        """Returns (hit int32 [rows,W], tuv float64 [rows,W,3]) as device tensors."""
        t = self.torch
        y1 = H if y1 is None else y1
        rows = y1 - y0
        hit = self._empty((rows, W), t.int32)
        tuv = t.zeros((rows, W, 3), dtype=t.float64, device=self.device)
        check(self.lib.rt_trace_primary(self.h, W, H, float(tan_half), y0, y1, hit.data_ptr(),
                                        tuv.data_ptr()), "rt_trace_primary")
        return hit, tuv

    def trace_rays(self, S, E, static: bool = False):
        """S, E: [n,3] float64 (numpy or device tensors).  Returns hit, tuv, point tensors.
        static=True goes through rt_trace_rays_static (the library allocates nothing)."""
        t = self.torch
        S = t.as_tensor(S, dtype=t.float64, device=self.device).contiguous()
        E = t.as_tensor(E, dtype=t.float64, device=self.device).contiguous()
        n = S.shape[0]
        hit = self._empty((n,), t.int32)
        tuv = t.zeros((n, 3), dtype=t.float64, device=self.device)
        point = t.zeros((n, 3), dtype=t.float64, device=self.device)
        fn = self.lib.rt_trace_rays_static if static else self.lib.rt_trace_rays
        check(fn(self.h, n, S.data_ptr(), E.data_ptr(), hit.data_ptr(), tuv.data_ptr(), point.data_ptr()),
              "rt_trace_rays_static" if static else "rt_trace_rays")
        return hit, tuv, point

    def occluded(self, S, E):
        t = self.torch
        S = t.as_tensor(S, dtype=t.float64, device=self.device).contiguous()
        E = t.as_tensor(E, dtype=t.float64, device=self.device).contiguous()
        n = S.shape[0]
        occ = self._empty((n,), t.uint8)
        check(self.lib.rt_occluded(self.h, n, S.data_ptr(), E.data_ptr(), occ.data_ptr()), "rt_occluded")
        return occ

    def render_rows(self, frame: RtFrame, W: int, H: int, tan_half: float, y0: int = 0,
                    y1: int | None = None, want=("hit", "tuv", "argb", "rgb8"), out=None,
                    rows_set: RtRows | None = None):
        """driver.c:126-193 for rows [y0,y1), or for an interleaved row set (rows_part).
        Returns a dict of device tensors indexed by local row."""
        t = self.torch
        y1 = H if y1 is None else y1
        rows = y1 - y0 if rows_set is None else rows_count(rows_set)
        o = {} if out is None else out
        if "hit" in want and "hit" not in o:
            o["hit"] = self._empty((rows, W), t.int32)
        if "tuv" in want and "tuv" not in o:
            o["tuv"] = t.zeros((rows, W, 3), dtype=t.float64, device=self.device)
        if "argb" in want and "argb" not in o:
            o["argb"] = self._empty((rows, W, 3), t.float64)
        if "rgb8" in want and "rgb8" not in o:
            o["rgb8"] = self._empty((rows, W, 3), t.uint8)
        ptr = lambda k: o[k].data_ptr() if k in o else None
        if rows_set is None:
            check(self.lib.rt_render_rows(self.h, C.byref(frame), W, H, float(tan_half), y0, y1,
                                          ptr("hit"), ptr("tuv"), ptr("argb"), ptr("rgb8")),
                  "rt_render_rows")
        elif rows > 0:
            check(self.lib.rt_render_rows_part(self.h, C.byref(frame), W, H, float(tan_half),
                                               C.byref(rows_set), ptr("hit"), ptr("tuv"), ptr("argb"),
                                               ptr("rgb8")), "rt_render_rows_part")
        return o

    def render_sequence(self, first_frame: int, count: int, W: int, H: int, tan_half: float,
                        frames_per_turn: int = 60, frame_step: int = 1,
                        bmp_pattern: str | None = None) -> RtSequenceStats:
        """driver.c:28-203: `count` frames of the orbit, pipelined; pictures written as 32-bit
        BMP to bmp_pattern % frame when a pattern is given."""
        st = RtSequenceStats()
        check(self.lib.rt_render_sequence(self.h, first_frame, count, frame_step, frames_per_turn, W, H,
                                          float(tan_half), bmp_pattern.encode() if bmp_pattern else None,
                                          C.byref(st)), "rt_render_sequence")
        return st

    def place_rows(self, rgb8, W: int, H: int, y0: int = 0, y1: int = 0, screen=None,
                   rows_set: RtRows | None = None):
        t = self.torch
        if screen is None:
            screen = t.zeros((H, W, 3), dtype=t.uint8, device=self.device)
        if rows_set is None:
            check(self.lib.rt_place_rows(rgb8.data_ptr(), W, H, y0, y1, screen.data_ptr()), "rt_place_rows")
        else:
            check(self.lib.rt_place_rows_part(rgb8.data_ptr(), W, H, C.byref(rows_set), screen.data_ptr()),
                  "rt_place_rows_part")
        return screen


class MultiScene:
    """One frame on several GPUs from this process (rt_multi in include/rt_amd.h): the mesh on
    every device, interleaved row parts, the rows gathered to the first device over RCCL."""

    def __init__(self, mesh: Mesh, devices):
        import torch

        self.torch = torch
        self.lib = _lib.load()
        devs = np.ascontiguousarray(list(devices), np.int32)
        h = C.c_void_p()
        check(self.lib.rt_multi_create(C.byref(h), len(devs), _ip(devs)), "rt_multi_create")
        self.h = h
        self.devices = [int(d) for d in devs]
        if isinstance(mesh, ObjFile):
            check(self.lib.rt_multi_set_obj(self.h, mesh.h), "rt_multi_set_obj")
            return
        errors = []

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
        def upload(scene, part, _user):
            try:
                Scene.adopt(scene, self.devices[part]).set_mesh(mesh)
                return 0
            except Exception as e:          # an exception must not cross the C frame
                errors.append(e)
                return -1

        rc = self.lib.rt_multi_for_each_scene(self.h, upload, None)
        if errors:
            raise errors[0]
        check(rc, "rt_multi_for_each_scene")

    @property
    def exchange(self) -> str:
        """How the rows reach the first device: "rccl", "copy" or "copy (<why not RCCL>)"."""
        return self.lib.rt_multi_exchange(self.h).decode()

    def transform(self, m) -> None:
        m = _mat(m)
        check(self.lib.rt_multi_transform(self.h, m.ctypes.data), "rt_multi_transform")

    def render_frame(self, frame: RtFrame, W: int, H: int, tan_half: float, screen=None):
        """The placed W x H x 3 picture on the first device."""
        t = self.torch
        if screen is None:
            screen = t.empty((H, W, 3), dtype=t.uint8, device=t.device("cuda", self.devices[0]))
        check(self.lib.rt_render_frame_multi(self.h, C.byref(frame), W, H, float(tan_half), screen.data_ptr()),
              "rt_render_frame_multi")
        return screen

    def close(self):
        if self.h:
            self.lib.rt_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_tile_culling(on) -> None:
    """rt_set_tile_culling: True / False, or None to follow RT_AMD_CULL again."""
    check(_lib.load().rt_set_tile_culling(-1 if on is None else (1 if on else 0)))


def tile_culling() -> bool:
    return bool(_lib.load().rt_get_tile_culling())


def profile_enable(on: bool = True) -> None:
    check(_lib.load().rt_profile_enable(1 if on else 0))


def profile_reset() -> None:
    check(_lib.load().rt_profile_reset())


def profile_get() -> RtProfile:
    p = RtProfile()
    check(_lib.load().rt_profile_get(C.byref(p)))
    return p


def save_bmp(path: str, screen_rgb: np.ndarray) -> None:
    a = np.ascontiguousarray(screen_rgb, dtype=np.uint8)
    H, W, _ = a.shape
    check(_lib.load().rt_save_bmp(path.encode(), a.ctypes.data, W, H), "rt_save_bmp")
