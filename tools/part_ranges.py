"""Prints every any-hit range of one interleaved row part of the bench frame with its time and
rate (development aid; RT_AMD_TRACE_PASSES with the profiling events on)."""
import os, sys
os.environ["RT_AMD_TRACE_PASSES"] = "1"
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
os.environ.pop("RT_AMD_TRACE_PASSES")
sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=("hit", "rgb8"))
torch.cuda.synchronize()
os.environ["RT_AMD_TRACE_PASSES"] = "1"
kernel.profile_enable(True)
sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=("hit", "rgb8"))
torch.cuda.synchronize()
