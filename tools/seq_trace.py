"""Renders a short frame sequence without output through rt_render_sequence (for rocprofv3 kernel
and memory-copy traces; development aid): seq_trace.py <workload> <frames>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import kernel
w, n = sys.argv[1], int(sys.argv[2])
kind, tris, W, H = bench.WORKLOADS[w]
sc = kernel.Scene(bench.make_mesh(kind, tris))
sc.render_sequence(0, 3, W, H, bench.TAN_HALF, 360)
st = sc.render_sequence(0, n, W, H, bench.TAN_HALF, 360)
print(f"{st.frames / st.seconds_total:.1f} frames/s")
