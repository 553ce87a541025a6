"""The boundary from C: tests/c/boundary_c11.c is a plain C11 translation unit that includes
include/rt_amd.h, links with -lrt_amd and calls the reference's two kernel.h:10-17 functions
the way driver.c:110,135 does.  Without a GPU: the header is valid C11 (layout asserts included)
and the program links against the library.  On the GPU: its printed results equal the oracle's
bit for bit."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from raytracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "boundary_c11.c")
EXE = os.path.join(ROOT, "tests", "c", "boundary_c11")


def test_header_is_valid_c11_and_program_links(tmp_path):
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    subprocess.check_call([gcc, "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), SRC])
    # C++ view of the same header, for the asserts' other spelling
    cxx = tmp_path / "inc.cpp"
    cxx.write_text('#include "rt_amd.h"\nint main() { return sizeof(Triangle) == 488 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(cxx)])
    out = tmp_path / "boundary_c11"
    subprocess.check_call([gcc, "-std=c11", "-I", os.path.join(ROOT, "include"), SRC, "-o", str(out),
                           "-L", os.path.join(ROOT, "raytracer_amd"), "-lrt_amd"])
    needed = subprocess.check_output(["readelf", "-d", str(out)], text=True)
    assert "librt_amd.so" in needed


@pytest.mark.gpu
def test_c_program_matches_oracle():
    assert os.path.exists(EXE), "tests/c/boundary_c11 is built by __graft_entry__.build()"
    text = subprocess.check_output([EXE], text=True, timeout=120)
    verts, rays = {}, {}
    for line in text.splitlines():
        f = line.split()
        if f[0] == "v":
            verts[int(f[1])] = [float.fromhex(t) for t in f[2:5]]
        elif f[0] == "r":
            rays[int(f[1])] = (int(f[2]), [float.fromhex(t) for t in f[3:8]])
    assert len(verts) == 7 and len(rays) == 3
    x = np.array([0, -1, 1, 0, -3, 3, 0], float)
    y = np.array([0, -1, -1, 1, -3, -3, 3], float)
    z = np.array([0, 5, 5, 5, 9, 9, 9], float)
    one = np.ones(2)
    mesh = scenes.Mesh(x, y, z, one, one, one, *[np.array(a, np.int32) for a in ([1, 4], [2, 5], [3, 6])],
                       *[np.array([1, 1], np.int32)] * 3)
    m = np.array([[1.5, 0, 0, 0.25], [0, 1.5, 0, -0.125], [0, 0, 2, 1], [0, 0, 0, 1]])
    X, Y, Z = O.transform(mesh, m)
    for k in range(7):
        assert verts[k] == [X[k], Y[k], Z[k]]
    osc = O.Scene(mesh, X, Y, Z)
    want_hits = []
    for r, E in enumerate([[0.02, 0.01, 1], [0.15, -0.1, 1], [5, 5, 1]]):
        idx, t, uv, p = O.intersect(osc, [0, 0, 0], E)
        want_hits.append(idx)
        got_idx, got = rays[r]
        assert got_idx == idx
        if idx == -1:
            assert got == [-7.0] * 5          # uv and point untouched on a miss
        else:
            assert got == [uv[0], uv[1], p[0], p[1], p[2]]
    assert want_hits == [0, 1, -1]
