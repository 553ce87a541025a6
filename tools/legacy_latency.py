"""Per-call latency of the reference's one-ray API, intersect_all_triangles_device
(kernel.h:16-17), with the mesh cached on the device (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import kernel, scenes

for tris in (6320, 100000, 1000000):
    mesh = scenes.torus(tris)
    ta = kernel.TriangleArray(mesh)
    x, y, z = (np.ascontiguousarray(a) for a in (mesh.x, mesh.y, mesh.z))
    uv, pt = np.zeros(2), np.zeros(3)
    rng = np.random.default_rng(1)
    S = np.array([0.0, 0.0, -30.0])
    n = 200
    kernel.intersect_all_triangles_device(S, np.zeros(3), uv, pt, ta, x, y, z, mesh.num_tris)   # upload
    t0 = time.perf_counter()
    hits = 0
    for i in range(n):
        E = S + np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 1.0])
        hits += kernel.intersect_all_triangles_device(S, E, uv, pt, ta, x, y, z, mesh.num_tris) >= 0
    dt = (time.perf_counter() - t0) / n
    print(f"{mesh.num_tris} triangles: {dt*1e6:.1f} us per ray ({hits}/{n} hit)", flush=True)
