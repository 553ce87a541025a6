"""Randomised stress of the fp32 filter's conservativeness: hit indices of the HIP path
against the CPU oracle on geometry chosen to be hard for fp32 (sub-pixel triangles, slivers,
grazing rays, rays starting on surfaces, far-away scenes, mixed scales).  Development aid;
prints mismatch counts per case."""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from raytracer_amd import kernel, scenes


def soup(n, seed, size, slivers=False, spread=7.5, offset=(0, 0, 0)):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-spread, spread, size=(n, 3)) + np.asarray(offset, float)
    s = size if np.isscalar(size) else rng.choice(size, size=(n, 1))
    e1 = rng.uniform(-1, 1, size=(n, 3)) * s
    e2 = rng.uniform(-1, 1, size=(n, 3)) * s
    if slivers:
        e2 = e1 * rng.uniform(0.2, 1.5, size=(n, 1)) + rng.uniform(-1, 1, size=(n, 3)) * s * 1e-4
    P = np.empty((3 * n + 1, 3)); P[0] = 0; P[1::3] = c; P[2::3] = c + e1; P[3::3] = c + e2
    A = np.arange(1, 3 * n + 1, 3, dtype=np.int32)
    nrm = np.vstack([np.zeros((1, 3)), np.cross(e1, e2)])
    idn = np.arange(1, n + 1, dtype=np.int32)
    return scenes.Mesh(x=P[:, 0].copy(), y=P[:, 1].copy(), z=P[:, 2].copy(), xn=nrm[:, 0].copy(),
                       yn=nrm[:, 1].copy(), zn=nrm[:, 2].copy(), A=A, B=A + 1, C=A + 2,
                       An=idn, Bn=idn.copy(), Cn=idn.copy())


def rays_for(mesh, kind, n, seed):
    rng = np.random.default_rng(seed + 1000)
    P = np.stack([mesh.x[1:], mesh.y[1:], mesh.z[1:]], 1)
    lo, hi = P.min(0), P.max(0)
    cen, ext = (lo + hi) / 2, (hi - lo).max()
    tri = rng.integers(0, mesh.num_tris, size=n)
    A = P[mesh.A[tri] - 1]; B = P[mesh.B[tri] - 1]; Cc = P[mesh.C[tri] - 1]
    if kind == "camera":            # common origin outside, aimed at random points of triangles
        S = np.tile(cen + np.array([0, 0, -2.0 * ext]), (n, 1))
        w = rng.dirichlet([1, 1, 1], size=n)
        T = w[:, :1] * A + w[:, 1:2] * B + w[:, 2:] * Cc
        return S, S + (T - S) * rng.uniform(0.3, 3.0, size=(n, 1))
    if kind == "edges":             # aimed exactly at edges / vertices
        w = rng.uniform(0, 1, size=(n, 1))
        T = np.where(rng.uniform(size=(n, 1)) < 0.2, A, w * A + (1 - w) * B)
        S = cen + rng.normal(size=(n, 3)) * ext
        return S, T
    if kind == "surface":           # shadow-like: start 0.001 off a triangle, go anywhere
        w = rng.dirichlet([1, 1, 1], size=n)
        p = w[:, :1] * A + w[:, 1:2] * B + w[:, 2:] * Cc
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        return p + 0.001 * d, cen + rng.normal(size=(n, 3)) * ext
    if kind == "grazing":           # nearly in the plane of the target triangle
        w = rng.dirichlet([1, 1, 1], size=n)
        p = w[:, :1] * A + w[:, 1:2] * B + w[:, 2:] * Cc
        nrm = np.cross(B - A, Cc - A); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-300
        tang = (B - A) / (np.linalg.norm(B - A, axis=1, keepdims=True) + 1e-300)
        d = tang + nrm * rng.uniform(-1e-3, 1e-3, size=(n, 1))
        return p - d * ext * rng.uniform(0.1, 1.0, size=(n, 1)), p + d * 1e-3
    raise ValueError(kind)


CASES = [
    ("soup r=0.5", lambda s: soup(3000, s, 0.5)),
    ("sub-pixel 1e-3", lambda s: soup(3000, s, 1e-3, spread=0.05)),
    ("slivers", lambda s: soup(3000, s, 0.5, slivers=True)),
    ("mixed scales", lambda s: soup(3000, s, [1e-3, 0.05, 1.0, 20.0])),
    ("far offset", lambda s: soup(3000, s, 0.3, offset=(300, -200, 500))),
    ("torus mesh", lambda s: scenes.torus(4000)),
]
KINDS = ("camera", "edges", "surface", "grazing")


def run(seeds, rays, verbose=True):
    """Returns (total mismatches, total rays, total oracle hits)."""
    total_bad = total_rays = total_hits = 0
    for name, make in CASES:
        for seed in range(seeds):
            mesh = make(seed)
            sc = kernel.Scene(mesh)
            sc.transform(np.eye(4))
            osc = O.Scene(mesh)
            for kind in KINDS:
                S, E = rays_for(mesh, kind, rays, seed)
                hit, tuv, _ = sc.trace_rays(S, E)
                occ = sc.occluded(S, E).cpu().numpy()
                hit = hit.cpu().numpy()
                rh, rt, _ = O.trace_rays(osc, S, E)
                bad = int((hit != rh).sum())
                badocc = int((occ != (rh >= 0)).sum())
                m = (hit == rh) & (rh >= 0)
                badt = int((tuv.cpu().numpy()[m] != rt[m]).any(axis=1).sum())
                total_bad += bad + badocc + badt
                total_rays += len(S)
                total_hits += int((rh >= 0).sum())
                if verbose:
                    print(f"{name:16s} seed {seed} {kind:8s} rays {len(S)} hits {(rh>=0).mean():.2f} "
                          f"mismatch hit {bad} occ {badocc} tuv {badt}", flush=True)
            sc.close()
    return total_bad, total_rays, total_hits


def run_frames(seeds, W=128, H=96, verbose=True):
    """Whole frames (primary sweep on the pixel grid + the shadow sweeps of every light +
    shading) of the same hard scenes against the oracle's pixel loop.  Beside the driver's two
    lights each frame gets one light placed 0.05 units off a random triangle and one far away.
    Returns (pixels with a different hit or t/u/v, pixels whose 8-bit colour is off by more
    than 1, pixels, hit pixels)."""
    import copy
    bad_hit = bad_rgb = pixels = hits = 0
    tan_half = math.tan(math.pi / 6)
    for name, make in CASES:
        for seed in range(seeds):
            mesh = copy.deepcopy(make(seed))
            scenes.center_and_scale(mesh)           # what the reference's loader does (obj_reader.c:239-289)
            fr = kernel.driver_frame(7 * seed + 3)
            obmat = kernel.frame_matrix(fr, "obmat")
            X, Y, Z = O.transform(mesh, obmat)
            rng = np.random.default_rng(seed + 77)
            t = int(rng.integers(0, mesh.num_tris))
            a = mesh.A[t]
            near = np.array([X[a], Y[a], Z[a]]) + 0.05 * rng.normal(size=3)
            extra = [near, np.array([400.0, -300.0, 250.0])]
            for k, pos in enumerate(extra):
                l = fr.num_lights
                for c in range(3):
                    fr.light_in_eye_space[l][c] = float(pos[c])
                    fr.light_rgb[l][c] = 0.6
                fr.light_attenuation[l][0] = 0.0
                fr.light_attenuation[l][1] = 0.0
                fr.light_max_dist[l] = 1e9
                fr.num_lights = l + 1
            sc = kernel.Scene(mesh)
            sc.transform(obmat)
            got = {k: v.cpu().numpy() for k, v in sc.render_rows(fr, W, H, tan_half).items()}
            ref = O.render_rows(O.Scene(mesh, X, Y, Z), O.frame_from(fr), W, H, tan_half)
            m = ref["hit"] >= 0
            bh = int((got["hit"] != ref["hit"]).sum()) + int((got["tuv"][m] != ref["tuv"][m]).any(axis=1).sum())
            br = int((np.abs(got["rgb8"].astype(int) - ref["rgb8"].astype(int)) > 1).any(axis=2).sum())
            bad_hit += bh; bad_rgb += br; pixels += W * H; hits += int(m.sum())
            if verbose:
                print(f"{name:16s} seed {seed} frame {W}x{H} lights {fr.num_lights} hit pixels {m.mean():.2f} "
                      f"shadow rays {ref['shadow_rays']} mismatch hit/tuv {bh} rgb8 {br}", flush=True)
            sc.close()
    return bad_hit, bad_rgb, pixels, hits


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--rays", type=int, default=40000)
    a = ap.parse_args()
    t00 = time.time()
    bad, n, h = run(a.seeds, a.rays)
    print(f"TOTAL mismatches {bad} of {n} rays ({h} hits)  ({time.time()-t00:.0f} s)")
    for W, H in ((128, 96), (1000, 40)):       # the second: long pixel rows for the sweep that steps along X
        bh, br, px, hp = run_frames(a.seeds, W, H)
        print(f"FRAMES {W}x{H}: hit/tuv mismatches {bh}, rgb8 off by more than 1: {br}, of {px} pixels ({hp} hit)  "
              f"({time.time()-t00:.0f} s)")
