"""Random image sizes, row parts, meshes and light counts: every part's hit ids against the CPU
oracle, RGB within 1/255, parts against the whole frame bit for bit (development aid)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as O
from raytracer_amd import kernel, scenes

TAN_HALF = np.tan(np.pi / 6)


def run(cases, seed, verbose=True):
    """Number of failing cases."""
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(cases):
        bad += not one_case(rng, case, verbose)
    return bad


def one_case(rng, case, verbose):
    W = int(rng.integers(1, 400)); H = int(rng.integers(1, 260))
    parts = int(rng.integers(1, 10)); block = int(rng.choice([1, 3, 8, 16, 16, 16, 32, 40]))
    tris = int(rng.choice([1, 7, 127, 128, 129, 500, 2500, 9000]))
    kind = rng.choice(["soup", "torus", "sphere"])
    frame_no = int(rng.integers(0, 60))
    mesh = {"soup": lambda: scenes.triangle_soup(tris, seed=case), "torus": lambda: scenes.torus(max(tris, 8)),
            "sphere": lambda: scenes.sphere(max(tris, 8))}[kind]()
    fr = kernel.driver_frame(frame_no)
    fr.num_lights = int(rng.integers(0, 3))
    obmat = kernel.frame_matrix(fr, "obmat")
    sc = kernel.Scene(mesh)
    sc.transform(obmat)
    whole = {k: v.cpu().numpy() for k, v in sc.render_rows(fr, W, H, TAN_HALF).items()}
    X, Y, Z = O.transform(mesh, obmat)
    ref = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, TAN_HALF)
    mism = int((whole["hit"] != ref["hit"]).sum())
    drgb = int(np.abs(whole["rgb8"].astype(int) - ref["rgb8"].astype(int)).max())
    part_bad = 0
    seen = np.zeros(H, int)
    for p in range(parts):
        rs = kernel.rows_part(H, parts, p, block)
        rows = kernel.rows_image_rows(rs)
        seen[rows] += 1
        got = sc.render_rows(fr, W, H, TAN_HALF, rows_set=rs)
        for k in ("hit", "argb", "rgb8"):
            if not np.array_equal(got[k].cpu().numpy(), whole[k][rows]):
                part_bad += 1
    ok = mism == 0 and drgb <= 1 and part_bad == 0 and np.all(seen == 1)
    if verbose or not ok:
        print(f"case {case}: {W}x{H} parts {parts} block {block} {kind} {mesh.num_tris} tris, {fr.num_lights} lights, "
              f"frame {frame_no}: hits {(ref['hit'] >= 0).sum()}, id mismatches {mism}, max |dRGB| {drgb}, "
              f"parts differing {part_bad} {'OK' if ok else 'FAIL'}", flush=True)
    sc.close()
    return ok


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    failures = run(a.cases, a.seed)
    print("failures:", failures)
    sys.exit(1 if failures else 0)
