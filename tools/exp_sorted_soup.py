"""Experiment (GPU box): how much of the primary launch's time is the scatter of a triangle order
that has nothing to do with the screen?  The 1 M-triangle soup is rendered as generated (random
order) and with its triangles renumbered along a Morton curve over the screen tile their centroid
projects to; prints the primary launch and frame times of both.  Development aid.
    python tools/exp_sorted_soup.py [tris] [W] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
import numpy as np
import torch
from raytracer_amd import kernel, scenes

TAN_HALF = math.tan(30 * math.pi / 180)
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160


def morton(ix, iy):
    def spread(v):
        v = v.astype(np.uint64) & 0xFFFF
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return spread(ix) | (spread(iy) << 1)


def run(mesh, label):
    frame = kernel.driver_frame(0)
    obmat = kernel.frame_matrix(frame, "obmat")
    sc = kernel.Scene(mesh, device=0)
    for _ in range(2):
        sc.transform(obmat)
        sc.render_rows(frame, W, H, TAN_HALF, want=("hit", "rgb8"))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sc.transform(obmat)
        r = sc.render_rows(frame, W, H, TAN_HALF, want=("hit", "rgb8"))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    kernel.profile_enable(True)
    kernel.profile_reset()
    sc.trace_primary(W, H, TAN_HALF)
    p = kernel.profile_get()
    kernel.profile_enable(False)
    print(f"{label}: frame {ms:.3f} ms, primary launch {p.sweep_ms:.3f} ms", flush=True)
    return r


mesh = scenes.triangle_soup(tris)
frame = kernel.driver_frame(0)
m = kernel.frame_matrix(frame, "obmat")
run(mesh, "file order")
cx = (mesh.x[mesh.A] + mesh.x[mesh.B] + mesh.x[mesh.C]) / 3
cy = (mesh.y[mesh.A] + mesh.y[mesh.B] + mesh.y[mesh.C]) / 3
cz = (mesh.z[mesh.A] + mesh.z[mesh.B] + mesh.z[mesh.C]) / 3
ex = m[0, 0] * cx + m[0, 1] * cy + m[0, 2] * cz + m[0, 3]
ey = m[1, 0] * cx + m[1, 1] * cy + m[1, 2] * cz + m[1, 3]
ez = m[2, 0] * cx + m[2, 1] * cy + m[2, 2] * cz + m[2, 3]
f = (W / 2) / TAN_HALF
px = np.where(ez > 1e-9, ex / np.maximum(ez, 1e-9) * f + W / 2, -1e9)
py = np.where(ez > 1e-9, ey / np.maximum(ez, 1e-9) * f + H / 2, -1e9)
for tile in (16, 64):
    ix = np.clip(px // tile, 0, 4095).astype(np.int64)
    iy = np.clip(py // tile, 0, 4095).astype(np.int64)
    order = np.argsort(morton(ix, iy), kind="stable")
    s = mesh.copy()
    for k in ("A", "B", "C", "An", "Bn", "Cn"):
        setattr(s, k, np.ascontiguousarray(getattr(mesh, k)[order]))
    run(s, f"Morton order over {tile}-pixel tiles")
# rows only: sorted by the screen row of the centroid
order = np.argsort(np.clip(py, -1, H), kind="stable")
s = mesh.copy()
for k in ("A", "B", "C", "An", "Bn", "Cn"):
    setattr(s, k, np.ascontiguousarray(getattr(mesh, k)[order]))
run(s, "sorted by screen row")
