"""Renders a few whole frames of a bench workload (for rocprofv3 kernel traces; development aid):
frame_trace.py <workload> <reps>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import kernel
w, reps = sys.argv[1], int(sys.argv[2])
kind, tris, W, H = bench.WORKLOADS[w]
mesh = bench.make_mesh(kind, tris)
sc = kernel.Scene(mesh)
out = None
for k in range(reps):
    fr = kernel.driver_frame(k, 360)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, bench.TAN_HALF, want=("hit", "rgb8"), out=out)
torch.cuda.synchronize()
