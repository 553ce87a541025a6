"""ctypes binding of librt_amd.so (the C ABI declared in include/rt_amd.h).

The library is the product; this module only declares argument types.  There is no CPU
fallback: importing works without a GPU (so symbol and host-logic tests can run), but every
device entry point returns RT_E_NODEVICE when no gfx950 is visible, and `load()` raises
when the shared object has not been built.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "librt_amd.so")   # env: kernel-variant A/B runs

RT_OK = 0
RT_MAX_LIGHTS = 100

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
Mat4 = (C.c_double * 4) * 4


class RtFrame(C.Structure):
    """rt_frame: light_model.h:4-8 globals + obmat/obinv of driver.c:106-107."""

    _fields_ = [
        ("num_lights", C.c_int),
        ("light_in_eye_space", (C.c_double * 3) * RT_MAX_LIGHTS),
        ("light_rgb", (C.c_double * 3) * RT_MAX_LIGHTS),
        ("light_attenuation", (C.c_double * 2) * RT_MAX_LIGHTS),
        ("light_max_dist", C.c_double * RT_MAX_LIGHTS),
        ("obmat", Mat4),
        ("obinv", Mat4),
    ]


class RtRows(C.Structure):
    """rt_rows: part `part` of `parts` of the rows [y0,y1), interleaved in blocks of `block` rows."""

    _fields_ = [("y0", C.c_int), ("y1", C.c_int), ("block", C.c_int), ("parts", C.c_int), ("part", C.c_int)]


class RtSequenceStats(C.Structure):
    _fields_ = [("frames", C.c_int), ("seconds_total", C.c_double), ("seconds_render", C.c_double),
                ("seconds_waiting_for_writers", C.c_double), ("seconds_writing", C.c_double)]


class RtProfile(C.Structure):
    _fields_ = [
        ("sweep_ms", C.c_double),
        ("sweep_launches", C.c_uint64),
        ("sweep_tests", C.c_uint64),
        ("shadow_ms", C.c_double),
        ("shadow_launches", C.c_uint64),
        ("shadow_tests", C.c_uint64),
        ("rays_primary", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("exact_candidates", C.c_uint64),
        ("exact_overflow", C.c_uint64),
        ("rays_live", C.c_uint64),
        ("cell_ms", C.c_double),
        ("cell_launches", C.c_uint64),
        ("cell_pairs", C.c_uint64),
        ("cell_ray_pairs", C.c_uint64),
        ("cell_rays", C.c_uint64),
        ("cells_per_lane", C.c_uint64),
        ("cell_lookups", C.c_uint64),
        ("cell_fine", C.c_uint64),
        ("cell_entries", C.c_uint64),
    ]


class RtObjView(C.Structure):
    _fields_ = (
        [(n, C.c_int) for n in ("num_v", "num_vn", "num_vt", "num_tris", "num_mats")]
        + [(n, c_double_p) for n in ("x", "y", "z", "xn", "yn", "zn", "u", "v")]
        + [(n, c_int_p) for n in ("A", "B", "C", "An", "Bn", "Cn", "At", "Bt", "Ct", "mtl")]
        + [(n, c_double_p) for n in ("Ka", "Kd", "Ks")]
        + [(n, c_int_p) for n in ("map_Ka", "map_Kd", "map_Ks")]
        + [("pinned", C.c_int)]
    )


# name -> (restype, argtypes); every symbol include/rt_amd.h declares
SIGNATURES = {
    "mat_mult_device": (None, [c_double_p] * 3 + [C.c_void_p] + [c_double_p] * 3 + [C.c_int]),
    "intersect_all_triangles_device": (
        C.c_int, [c_double_p] * 4 + [C.c_void_p] + [c_double_p] * 3 + [C.c_int]),
    "rt_legacy_invalidate": (None, []),
    "rt_set_triangle_stride": (None, [C.c_int]),
    "rt_init": (C.c_int, [C.c_int]),
    "rt_shutdown": (None, []),
    "rt_device_count": (C.c_int, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_version": (C.c_char_p, []),
    "rt_set_stream": (C.c_int, [C.c_void_p]),
    "rt_scene_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (C.c_int, [C.c_void_p]),
    "rt_scene_set_vertices": (C.c_int, [C.c_void_p, C.c_int] + [c_double_p] * 3),
    "rt_scene_set_normals": (C.c_int, [C.c_void_p, C.c_int] + [c_double_p] * 3),
    "rt_scene_set_texcoords": (C.c_int, [C.c_void_p, C.c_int] + [c_double_p] * 2),
    "rt_scene_set_triangles": (C.c_int, [C.c_void_p, C.c_int] + [c_int_p] * 10),
    "rt_scene_set_triangles_aos": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rt_scene_set_materials": (C.c_int, [C.c_void_p, C.c_int] + [c_double_p] * 3 + [c_int_p] * 3),
    "rt_scene_set_textures": (C.c_int, [C.c_void_p, C.c_int, c_int_p, c_int_p, C.POINTER(C.c_void_p)]),
    "rt_scene_transform": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_scene_bake": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_scene_get_eye_vertices": (C.c_int, [C.c_void_p] + [c_double_p] * 3),
    "rt_scene_num_tris": (C.c_int, [C.c_void_p]),
    "rt_trace_primary": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]),
    "rt_trace_rays": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5),
    "rt_trace_rays_static": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5),
    "rt_occluded": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 3),
    "rt_render_rows": (C.c_int, [C.c_void_p, C.POINTER(RtFrame), C.c_int, C.c_int, C.c_double,
                                 C.c_int, C.c_int] + [C.c_void_p] * 4),
    "rt_place_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rt_band": (C.c_int, [C.c_int, C.c_int, C.c_int, c_int_p, c_int_p]),
    "rt_rows_count": (C.c_int, [C.POINTER(RtRows)]),
    "rt_rows_image_row": (C.c_int, [C.POINTER(RtRows), C.c_int]),
    "rt_trace_primary_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(RtRows),
                                        C.c_void_p, C.c_void_p]),
    "rt_render_rows_part": (C.c_int, [C.c_void_p, C.POINTER(RtFrame), C.c_int, C.c_int, C.c_double,
                                      C.POINTER(RtRows)] + [C.c_void_p] * 4),
    "rt_place_rows_part": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(RtRows), C.c_void_p]),
    "rt_multi_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, c_int_p]),
    "rt_multi_destroy": (C.c_int, [C.c_void_p]),
    "rt_multi_num_devices": (C.c_int, [C.c_void_p]),
    "rt_multi_exchange": (C.c_char_p, [C.c_void_p]),
    "rt_multi_for_each_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_multi_set_obj": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_multi_transform": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_render_frame_multi": (C.c_int, [C.c_void_p, C.POINTER(RtFrame), C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "rt_render_sequence": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_char_p, C.POINTER(RtSequenceStats)]),
    "rt_sweep_config": (C.c_int, [c_int_p] * 4),
    "rt_apex_sweep_config": (C.c_int, [c_int_p, c_int_p, C.POINTER(C.c_double), c_int_p,
                                       C.POINTER(C.c_double), c_int_p]),
    "rt_raster_sweep_mix": (C.c_int, [c_int_p] * 5),
    "rt_raster_sweep_form": (C.c_int, [c_int_p]),
    "rt_raster_shape_query": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p,
                                        C.POINTER(C.c_long)]),
    "rt_set_tile_culling": (C.c_int, [C.c_int]),
    "rt_get_tile_culling": (C.c_int, []),
    "rt_profile_enable": (C.c_int, [C.c_int]),
    "rt_profile_reset": (C.c_int, []),
    "rt_profile_get": (C.c_int, [C.POINTER(RtProfile)]),
    "rt_driver_frame_setup": (C.c_int, [C.c_int, C.c_int, C.POINTER(RtFrame)]),
    "rt_view": (C.c_int, [C.c_void_p, C.c_void_p] + [c_double_p] * 3),
    "rt_movement_sequence": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_int_p, c_double_p]),
    "rt_mat_mult": (C.c_int, [C.c_void_p] * 3),
    "rt_mat_mult_pt": (C.c_int, [c_double_p, C.c_void_p, c_double_p]),
    "rt_obj_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rt_obj_free": (C.c_int, [C.c_void_p]),
    "rt_obj_get": (C.c_int, [C.c_void_p, C.POINTER(RtObjView)]),
    "rt_obj_num_textures": (C.c_int, [C.c_void_p]),
    "rt_obj_texture": (C.c_int, [C.c_void_p, C.c_int, c_int_p, c_int_p, C.POINTER(C.c_void_p)]),
    "rt_scene_set_obj": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_save_bmp": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
}

_lib = None


class RtError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load librt_amd.so and attach the signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(
            f"{LIB_PATH} is missing: build it with raytracer_amd/csrc/build.sh "
            "(or __graft_entry__.build()); there is no fallback path")
    # PyTorch ships a HIP runtime of its own under the same soname as the system one this
    # library is linked against.  Whichever is loaded first serves the whole process, and
    # torch cannot find the GPU through the system runtime ("No HIP GPUs are available"), so
    # torch's must come first wherever torch is going to own the device buffers.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != RT_OK:
        msg = load().rt_last_error().decode("utf-8", "replace")
        raise RtError(f"{what or 'rt call'} failed ({rc}): {msg}")
