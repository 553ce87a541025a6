"""Full-frame golden vectors at the BASELINE sizes the CPU oracle cannot sweep inside a test run:
the whole 1920x1080 frame of a 20 000-triangle mesh with materials (configuration 2's shape:
a named mesh of that class at 1080p on one GPU, hit ids bit-exact against the CPU), and the whole
3840x2160 frame of the 69 451-triangle mesh (configuration 3), hit ids only.  Produced by the
oracle (oracle/rt_oracle.c) in the build container, minutes of CPU time each, and committed as
compressed arrays: the hit index of every pixel, and the 8-bit colours of every 8th row.

    python tests/golden/make_fullframe_golden.py [config2|config3 ...]
"""
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O          # noqa: E402
from raytracer_amd import scenes        # noqa: E402

TAN_HALF = math.tan(30 * math.pi / 180)

CASES = {
    # name: (mesh factory, frame of the 60-frame orbit, W, H, shade)
    "config2_sphere20k_1080p_f3": (lambda: scenes.with_materials(scenes.sphere(20000), num_mats=4, seed=11), 3, 1920, 1080, True),
    "config3_torus69451_4k_f7": (lambda: scenes.torus(69451), 7, 3840, 2160, False),
}


def main():
    names = sys.argv[1:] or list(CASES)
    for name in names:
        key = [k for k in CASES if k.startswith(name)][0]
        make, frame_no, W, H, shade = CASES[key]
        mesh = make()
        fr = O.frame_setup(frame_no, 60)
        X, Y, Z = O.transform(mesh, O.arr4(fr.obmat))
        t0 = time.time()
        r = O.render_rows(O.Scene(mesh, X, Y, Z), fr, W, H, TAN_HALF, shade=shade)
        hit = r["hit"]
        # int32 hit ids compress well as differences along a row
        out = {"hit": hit, "triangles": mesh.num_tris}
        if shade:
            out["rgb8_rows"] = np.arange(4, H, 8)
            out["rgb8"] = r["rgb8"][4::8]
        np.savez_compressed(os.path.join(HERE, key + ".npz"), **out)
        print(f"{key}: {int((hit >= 0).sum())} hit pixels of {W * H}, {time.time() - t0:.0f} s, "
              f"{os.path.getsize(os.path.join(HERE, key + '.npz')) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
