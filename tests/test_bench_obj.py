"""bench.py --obj PATH: a user-supplied Wavefront file through the library's own loader
(rt_obj_load + rt_scene_set_obj) instead of the workload's procedural mesh; a path that does not
exist prints "asset not available" and keeps the stand-in.  The file here is teapot-style --
quads in face format 1 (`f v v v v`), the case the reference's loader stores pointers for instead
of indices (obj_reader.c:98-100); the loader implements the evident intent.

Host side: the mesh the bench would render equals the Python restatement of obj_reader.c.
GPU: the bench's line names the file, and a frame of that file equals the oracle's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench
from oracle import obj_oracle
from raytracer_amd import kernel, scenes
from conftest import TAN_HALF


def _write_quads(tmp_path, tris=600):
    mesh = scenes.torus(tris, normalize=False)
    path = tmp_path / "teapot_style.obj"
    text = scenes.to_obj_text(mesh, fmt=1, quads=True)
    assert text.count("\nf ") + text.startswith("f ") > 0 and all(len(l.split()) == 5 for l in text.splitlines() if l.startswith("f "))
    path.write_text(text)
    return str(path), sum(1 for l in text.splitlines() if l.startswith("f "))


def test_bench_mesh_path_loads_a_format_1_quad_file(tmp_path):
    path, quads = _write_quads(tmp_path)
    mesh, obj, note = bench.load_mesh("teapot_class_480p", path)
    assert obj is not None and obj.num_tris == 2 * quads == mesh.num_tris
    assert "teapot_style.obj" in note and "teapot_style.obj" in mesh.name and str(2 * quads) in mesh.name
    ref = obj_oracle.load(path)
    for k in ("x", "y", "z", "xn", "yn", "zn", "A", "B", "C", "An", "Bn", "Cn", "mtl"):
        assert np.array_equal(getattr(mesh, k), ref[k]), k
    # the quad split of obj_reader.c:167-231: (a, b, c) then (a, c, d)
    assert np.array_equal(mesh.A[0::2], mesh.A[1::2]) and np.array_equal(mesh.C[0::2], mesh.B[1::2])
    obj.close()


def test_missing_asset_keeps_the_stand_in(tmp_path, capsys):
    mesh, obj, note = bench.load_mesh("teapot_class_480p", str(tmp_path / "bunny.obj"))
    kind, tris, _, _ = bench.WORKLOADS["teapot_class_480p"]
    assert obj is None and mesh.num_tris == bench.make_mesh(kind, tris).num_tris      # (a tessellation near the target)
    assert "asset not available" in capsys.readouterr().err and "asset not available" in note
    mesh2, obj2, note2 = bench.load_mesh("teapot_class_480p", None)
    assert obj2 is None and note2 == "procedural" and mesh2.num_tris == mesh.num_tris


@pytest.mark.gpu
def test_bench_line_and_frame_of_the_loaded_file(tmp_path):
    from oracle import oracle as O
    path, quads = _write_quads(tmp_path, tris=2000)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "teapot_class_480p", "--obj", path,
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True,
                       timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["triangles"] == 2 * quads and "teapot_style.obj" in line["config"]["mesh"]
    assert "rt_obj_load" in line["config"]["mesh_source"] and line["value"] > 0
    assert line["comm"]["world_size"] == 1
    assert line["roofline"]["general_mt"]["hit_equals_pixel_grid_sweep"] is True
    # the same file, one frame, against the oracle fed by the Python restatement of the parser
    obj = kernel.ObjFile(path)
    ref = obj_oracle.load(path)
    mesh = scenes.Mesh(**{k: np.ascontiguousarray(ref[k]) for k in
                          ("x", "y", "z", "xn", "yn", "zn", "A", "B", "C", "An", "Bn", "Cn")},
                       mtl=np.ascontiguousarray(ref["mtl"]))
    fr = kernel.driver_frame(5)
    W, H = 160, 120
    X, Y, Z = O.transform(mesh, kernel.frame_matrix(fr, "obmat"))
    want = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF)
    sc = kernel.Scene(obj)
    sc.transform(kernel.frame_matrix(fr, "obmat"))
    out = sc.render_rows(fr, W, H, TAN_HALF)
    assert (want["hit"] >= 0).sum() > 300
    assert np.array_equal(out["hit"].cpu().numpy(), want["hit"])
    assert np.abs(out["rgb8"].cpu().numpy().astype(int) - want["rgb8"].astype(int)).max() <= 1
    sc.close()
    obj.close()
