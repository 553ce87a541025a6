"""Times the primary sweep of a bench workload for several forced lane widths and chunk counts
(development aid): raster_shapes.py [workload] [shape ...], a shape being "<pixels per lane>,<chunks>"."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from raytracer_amd import kernel
wl = sys.argv[1] if len(sys.argv) > 1 else "soup1m_4k"
shapes = sys.argv[2:] or ["64,61", "128,61", "256,61"]
kind, tris, W, H = bench.WORKLOADS[wl]
mesh = bench.make_mesh(kind, tris); fr = kernel.driver_frame(0)
sc = kernel.Scene(mesh); sc.transform(kernel.frame_matrix(fr, "obmat"))
for force in [None] + shapes:
    if force: os.environ["RT_AMD_RASTER_FORCE"] = force
    for i in range(3):
        kernel.profile_enable(True); kernel.profile_reset()
        sc.render_rows(fr, W, H, bench.TAN_HALF, want=("rgb8",)); torch.cuda.synchronize()
        p = kernel.profile_get(); kernel.profile_enable(False)
    print(wl, force, "primary ms", round(p.sweep_ms, 2), "%.3e tests/s" % (p.sweep_tests / p.sweep_ms * 1e3), "shadow ms", round(p.shadow_ms, 2), flush=True)
