"""Primary-sweep time of one interleaved row part of the bench frame for a list of forced launch
shapes (development aid): shape_sweep.py <parts> <part> "<pixels per lane>,<chunks>[,<waves side
by side>]" ...   ("auto" = the launcher's own choice)"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import kernel, _lib

parts, part = int(sys.argv[1]), int(sys.argv[2])
shapes = sys.argv[3:] or ["auto"]
kind, tris, W, H = bench.WORKLOADS[os.environ.get("WORKLOAD", "soup1m_4k")]
mesh = bench.make_mesh(kind, tris)
fr = kernel.driver_frame(0)
sc = kernel.Scene(mesh)
sc.transform(kernel.frame_matrix(fr, "obmat"))
rs = kernel.rows_part(H, parts, part)
rows = kernel.rows_count(rs)
hit = torch.empty((rows, W), dtype=torch.int32, device="cuda")
lib = _lib.load()
kernel.profile_enable(True)
ref = None
for shape in shapes:
    if shape == "auto":
        os.environ.pop("RT_AMD_RASTER_FORCE", None)
    else:
        os.environ["RT_AMD_RASTER_FORCE"] = shape
    best = None
    for rep in range(3):
        kernel.profile_reset()
        _lib.check(lib.rt_trace_primary_part(sc.h, W, H, bench.TAN_HALF, C.byref(rs), hit.data_ptr(), None))
        p = kernel.profile_get()
        best = p.sweep_ms if best is None else min(best, p.sweep_ms)
    if ref is None:
        ref = hit.clone()
    same = bool(torch.equal(ref, hit))
    tile, ppl, spl, chunks = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    f1, f2 = C.c_double(), C.c_double()
    lib.rt_apex_sweep_config(C.byref(tile), C.byref(ppl), C.byref(f1), C.byref(spl), C.byref(f2), C.byref(chunks))
    print(f"{shape:>12}: {best:8.3f} ms  {p.sweep_tests / best / 1e9:6.2f} T/s  (ran {ppl.value} px/lane, {chunks.value} chunks; "
          f"{p.exact_candidates} candidates, {p.exact_overflow} overflows, same hits {same})", flush=True)
