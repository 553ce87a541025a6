"""Predicts strong scaling on one GPU: renders each of the n row parts of the bench frame alone
(interleaved 16-row blocks, or contiguous bands with --bands) and compares the slowest part with
the whole frame (development aid)."""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import kernel

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="soup1m_4k")
ap.add_argument("--ranks", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--detail", action="store_true")
ap.add_argument("--bands", action="store_true")
ap.add_argument("--block", type=int, default=16)
a = ap.parse_args()
kind, tris, W, H = bench.WORKLOADS[a.workload]
mesh = bench.make_mesh(kind, tris)
fr = kernel.driver_frame(0)
sc = kernel.Scene(mesh)
sc.transform(kernel.frame_matrix(fr, "obmat"))

def timed(rs):
    want = ("hit", "rgb8")
    sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=want)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=want)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if a.detail:
        kernel.profile_enable(True); kernel.profile_reset()
        sc.render_rows(fr, W, H, bench.TAN_HALF, rows_set=rs, want=want)
        p = kernel.profile_get(); kernel.profile_enable(False)
        print(f"   part {rs.part}/{rs.parts}: {dt*1e3:.1f} ms; primary {p.sweep_ms:.1f} ms ({p.sweep_tests/p.sweep_ms/1e9:.2f} T/s), "
              f"shadow {p.shadow_ms:.1f} ms in {p.shadow_launches} launches ({p.shadow_tests/max(p.shadow_ms,1e-9)/1e9:.2f} T/s, "
              f"{p.shadow_tests:.3e} tests, {p.rays_shadow} rays)", flush=True)
    return dt

full = timed(kernel.rows_part(H))
print(f"full frame {full*1e3:.1f} ms", flush=True)
for n in a.ranks:
    ts = []
    for r in range(n):
        if a.bands:
            y0, y1 = kernel.band(H, n, r)
            ts.append(timed(kernel.rows_part(H, 1, 0, y1 - y0, y0, y1)))
        else:
            ts.append(timed(kernel.rows_part(H, n, r, a.block)))
    print(f"{n} parts: max {max(ts)*1e3:.1f} ms sum {sum(ts)*1e3:.1f} ms -> predicted speed-up {full/max(ts):.2f}x "
          f"(bands: {' '.join(f'{t*1e3:.0f}' for t in ts)})", flush=True)
