"""Times the primary and shadow sweeps of the bench frame with tile culling on for several forced
lane widths of the primary sweep (development aid)."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from raytracer_amd import kernel
kind, tris, W, H = bench.WORKLOADS["soup1m_4k"]
mesh = bench.make_mesh(kind, tris); fr = kernel.driver_frame(0)
sc = kernel.Scene(mesh); sc.transform(kernel.frame_matrix(fr, "obmat"))
kernel.set_tile_culling(1) if hasattr(kernel, "set_tile_culling") else os.environ.__setitem__("RT_AMD_CULL", "1")
for force in (None, "64,61", "32,31", "16,31", "8,31"):
    if force: os.environ["RT_AMD_RASTER_FORCE"] = force
    for i in range(2):
        kernel.profile_enable(True); kernel.profile_reset()
        sc.render_rows(fr, W, H, bench.TAN_HALF, want=("rgb8",)); torch.cuda.synchronize()
        p = kernel.profile_get(); kernel.profile_enable(False)
    print(force, "primary ms", round(p.sweep_ms, 2), "shadow ms", round(p.shadow_ms, 2), flush=True)
