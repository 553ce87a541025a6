"""The general Moller-Trumbore sweep (rt_trace_rays) on the 4K frame's primary rays against a
bench workload's mesh (development aid): general_mt.py [workload]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import kernel
w = sys.argv[1] if len(sys.argv) > 1 else "mesh100k_4k"
kind, tris, W, H = bench.WORKLOADS[w]
sc = kernel.Scene(bench.make_mesh(kind, tris))
fr = kernel.driver_frame(0)
sc.transform(kernel.frame_matrix(fr, "obmat"))
hit, _ = sc.trace_primary(W, H, bench.TAN_HALF)
print(json.dumps(bench.general_mt(sc, W, H, torch.device("cuda", 0), hit)))
