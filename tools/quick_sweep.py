"""Quick timing of the closest-hit sweep on a synthetic mesh (development aid)."""
import argparse, math, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import kernel, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=100000)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mesh", default="soup")
a = ap.parse_args()
mesh = {"soup": scenes.triangle_soup, "torus": scenes.torus, "sphere": scenes.sphere}[a.mesh](a.tris)
fr = kernel.driver_frame(0)
sc = kernel.Scene(mesh)
sc.transform(kernel.frame_matrix(fr, "obmat"))
th = math.tan(math.pi / 6)
kernel.profile_enable(True)
for r in range(a.reps):
    kernel.profile_reset()
    t0 = time.time()
    hit, tuv = sc.trace_primary(a.w, a.h, th)
    torch.cuda.synchronize()
    dt = time.time() - t0
    p = kernel.profile_get()
    print(f"rep {r}: wall {dt*1e3:.1f} ms sweep {p.sweep_ms:.2f} ms tests {p.sweep_tests:.3e} "
          f"-> {p.sweep_tests/p.sweep_ms/1e9:.3f} Ttests/s = {46*p.sweep_tests/p.sweep_ms/1e9:.1f} TFLOP/s(46) "
          f"cand {p.exact_candidates} ovf {p.exact_overflow} hitfrac {(hit>=0).float().mean().item():.3f}", flush=True)
