"""The primary sweep's rare path finds the columns of a lane that lie in a triangle's bounding
box as an interval of column numbers (two subtractions) instead of evaluating every column of the
lane again -- and, in the form of the kernel that steps from row to row by a constant, the rows
of every 16-row group the same way.  A checking build of the same sources (raytracer_amd/librt_amd_check.so,
-DRT_RASTER_CHECK_INTERVAL=1) does both and counts the columns the hot loop's own test passes
that lie outside the interval; there must be none, on hard scenes and for every lane width."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK_LIB = os.path.join(ROOT, "raytracer_amd", "librt_amd_check.so")

SCRIPT = r"""
import math, sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
from raytracer_amd import kernel, scenes
import stress_parity as sp
tan_half = math.tan(math.pi / 6)
meshes = [scenes.triangle_soup(30000, seed=3), scenes.torus(20000)] + [make(1) for _, make in sp.CASES]
for mesh in meshes:
    scenes.center_and_scale(mesh)
    for frame in (0, 17):
        fr = kernel.driver_frame(frame)
        sc = kernel.Scene(mesh)
        sc.transform(kernel.frame_matrix(fr, "obmat"))
        sc.trace_primary(1280, 352, tan_half)
        sc.close()
"""


@pytest.mark.parametrize("shape", [None, "256,3,0", "512,2,0", "1024,3,0", "2048,2,0", "4096,2,0", "4096,5,-1", "8192,3,-1", "16384,4,-1", "16384,2,0", "32768,3", "65536,2", "147456,4", "512,3,-1", "128,4,1", "64,4,1", "32,4,2",
                                   "16,3,4", "8,2,1", "32,5,0", "2048,2,0/0", "256,3,0/0", "16,3,4/0"])
def test_no_pass_outside_the_interval(shape):
    if not os.path.exists(CHECK_LIB):
        pytest.fail("raytracer_amd/librt_amd_check.so is missing: run __graft_entry__.build()")
    env = dict(os.environ, RT_AMD_LIB=CHECK_LIB)
    if shape and shape.endswith("/0"):          # the form with a coordinate per row
        shape = shape[:-2]
        env["RT_AMD_RASTER_REGULAR"] = "0"
    if shape:
        env["RT_AMD_RASTER_FORCE"] = shape
    p = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, os.path.join(ROOT, "tools"))], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    counts = [int(m) for m in re.findall(r"interval check: (\d+) passes", p.stderr)]
    assert len(counts) >= 16, p.stderr[-2000:]
    assert sum(counts) == 0, counts
