"""Frame-sequence throughput (config 5: 6 k-triangle mesh, 4K orbit): rt_render_sequence with and
without BMP output, against the same frames rendered one call at a time (development aid)."""
import argparse, math, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import kernel

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=120)
ap.add_argument("--workload", default="teapot_seq_4k")
a = ap.parse_args()
kind, tris, W, H = bench.WORKLOADS[a.workload]
sc = kernel.Scene(bench.make_mesh(kind, tris))
sc.render_sequence(0, 3, W, H, bench.TAN_HALF, 360)          # warm-up: allocations
st = sc.render_sequence(0, a.frames, W, H, bench.TAN_HALF, 360)
print(f"no output   : {st.frames / st.seconds_total:7.1f} frames/s ({st.seconds_total * 1e3 / st.frames:.2f} ms/frame)")
d = tempfile.mkdtemp(prefix="rtseq_", dir="/tmp")
try:
    st = sc.render_sequence(0, a.frames, W, H, bench.TAN_HALF, 360, bmp_pattern=os.path.join(d, "pic%04d.bmp"))
    print(f"BMP to /tmp : {st.frames / st.seconds_total:7.1f} frames/s; render {st.seconds_render:.2f} s, "
          f"stalled on writers {st.seconds_waiting_for_writers:.2f} s, writing {st.seconds_writing:.2f} s "
          f"(4 threads), total {st.seconds_total:.2f} s, {len(os.listdir(d))} files")
    # unpipelined: copy and write inside the loop
    t0 = time.perf_counter()
    n = min(a.frames, 30)
    for k in range(n):
        fr = kernel.driver_frame(k, 360)
        sc.transform(kernel.frame_matrix(fr, "obmat"))
        out = sc.render_rows(fr, W, H, bench.TAN_HALF, want=("rgb8",))
        screen = sc.place_rows(out["rgb8"], W, H, 0, H)
        kernel.save_bmp(os.path.join(d, "ref%04d.bmp" % k), screen.cpu().numpy())
    dt = time.perf_counter() - t0
    print(f"one by one  : {n / dt:7.1f} frames/s (render, copy, write in turn)")
finally:
    shutil.rmtree(d, ignore_errors=True)
