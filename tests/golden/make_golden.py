"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle.

The reference holds no test vectors, images or meshes, so these fixtures are produced by
the oracle (oracle/rt_oracle.c, oracle/obj_oracle.py) in the build container and committed:
  *.obj + *.expected.json  -- loader fixtures (arrays after centring and scaling)
  frame_*.npz              -- small frames: hit index, t/u/v, argb, rgb8 for a seeded scene
Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import obj_oracle, oracle as O          # noqa: E402
from raytracer_amd import scenes                    # noqa: E402

TAN_HALF = math.tan(30 * math.pi / 180)

OBJS = {
    "quad_v": "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n",
    "quad_v_vt_vn": "v 0 0 0\nv 2 0 0\nv 2 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                    "vn 0 0 1\nvn 0 1 0\nvn 1 0 0\nvn 0 0 -1\nf 1/1/1 2/2/2 3/3/3 4/4/4\nf 1/1/4 3/3/2 2/2/1\n",
    "pyramid_v_vn": "# pyramid\nv -1 0 -1\nv 1 0 -1\nv 1 0 1\nv -1 0 1\nv 0 2 0\n"
                    "vn 0 -1 0\nvn 0 0.4472 -0.8944\nvn 0.8944 0.4472 0\nvn 0 0.4472 0.8944\nvn -0.8944 0.4472 0\n"
                    "f 1//1 2//1 3//1 4//1\nf 1//2 5//2 2//2\nf 2//3 5//3 3//3\nf 3//4 5//4 4//4\nf 4//5 5//5 1//5\n",
}


def main():
    for name, text in OBJS.items():
        p = os.path.join(HERE, name + ".obj")
        open(p, "w").write(text)
        r = obj_oracle.load(p)
        exp = {k: np.asarray(r[k]).tolist() for k in
               ("x", "y", "z", "xn", "yn", "zn", "A", "B", "C", "An", "Bn", "Cn", "mtl")}
        json.dump(exp, open(os.path.join(HERE, name + ".expected.json"), "w"))
    cases = {
        "frame_soup400_64x48_f3": (scenes.triangle_soup(400, seed=21), 3, 64, 48),
        "frame_torus600_80x60_f11": (scenes.with_materials(scenes.torus(600), num_mats=3, seed=5), 11, 80, 60),
        "frame_sphere300_48x64_f40": (scenes.sphere(300), 40, 48, 64),
    }
    for name, (mesh, frame_no, W, H) in cases.items():
        fr = O.frame_setup(frame_no, 60)
        X, Y, Z = O.transform(mesh, O.arr4(fr.obmat))
        r = O.render_rows(O.Scene(mesh, X, Y, Z), fr, W, H, TAN_HALF)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), hit=r["hit"], tuv=r["tuv"],
                            argb=r["argb"], rgb8=r["rgb8"], shadow_rays=r["shadow_rays"])
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
