"""Times one interleaved row part of the bench frame (development aid): part_time.py <parts> <part>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import kernel
parts, part = int(sys.argv[1]), int(sys.argv[2])
kind, tris, W, H = bench.WORKLOADS["soup1m_4k"]
mesh = bench.make_mesh(kind, tris)
fr = kernel.driver_frame(0)
sc = kernel.Scene(mesh)
sc.transform(kernel.frame_matrix(fr, "obmat"))
rs = kernel.rows_part(H, parts, part)
want = ("hit", "rgb8")
sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=want)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=want)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
kernel.profile_enable(True); kernel.profile_reset()
sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=want)
p = kernel.profile_get()
print(f"part {part}/{parts}: {min(ts)*1e3:.1f} ms; primary {p.sweep_ms:.1f} ms, shadow {p.shadow_ms:.1f} ms in "
      f"{p.shadow_launches} launches, {p.shadow_tests:.3e} tests ({p.shadow_tests/max(p.shadow_ms,1e-9)/1e9:.2f} T/s)", flush=True)
