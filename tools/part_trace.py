"""Renders one interleaved row part of the bench frame a few times (for rocprofv3 kernel traces)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import kernel
parts, part, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
block = int(sys.argv[4]) if len(sys.argv) > 4 else 16
kind, tris, W, H = bench.WORKLOADS["soup1m_4k"]
mesh = bench.make_mesh(kind, tris)
fr = kernel.driver_frame(0)
sc = kernel.Scene(mesh)
sc.transform(kernel.frame_matrix(fr, "obmat"))
rs = kernel.rows_part(H, parts, part, block)
for _ in range(reps):
    sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=("hit", "rgb8"))
torch.cuda.synchronize()
