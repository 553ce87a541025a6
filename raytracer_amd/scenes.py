"""Synthetic scenes for tests and benchmarks (the reference ships no meshes, and none can be
fetched): seeded random triangle soups and procedural closed meshes, as structure-of-arrays
numpy buffers laid out like the reference's globals (obj_reader.h:41-46: 1-indexed vertex,
normal and texture-coordinate arrays; per-triangle index arrays).
"""
from __future__ import annotations

from dataclasses import dataclass, field
import io

import numpy as np


@dataclass
class Mesh:
    """Host-side mesh.  x/y/z, xn/yn/zn, u/v have length count+1; slot 0 is unused."""

    x: np.ndarray
    y: np.ndarray
    z: np.ndarray
    xn: np.ndarray
    yn: np.ndarray
    zn: np.ndarray
    A: np.ndarray
    B: np.ndarray
    C: np.ndarray
    An: np.ndarray
    Bn: np.ndarray
    Cn: np.ndarray
    u: np.ndarray = field(default_factory=lambda: np.zeros(1))
    v: np.ndarray = field(default_factory=lambda: np.zeros(1))
    At: np.ndarray | None = None
    Bt: np.ndarray | None = None
    Ct: np.ndarray | None = None
    mtl: np.ndarray | None = None            # material index per triangle, -1 = none
    Ka: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    Kd: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    Ks: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    map_Ka: np.ndarray | None = None
    map_Kd: np.ndarray | None = None
    map_Ks: np.ndarray | None = None
    textures: list = field(default_factory=list)     # (h, w, 3) uint8 arrays, top row first
    name: str = "mesh"

    @property
    def num_v(self) -> int:
        return len(self.x) - 1

    @property
    def num_vn(self) -> int:
        return len(self.xn) - 1

    @property
    def num_vt(self) -> int:
        return len(self.u) - 1

    @property
    def num_tris(self) -> int:
        return len(self.A)

    def copy(self) -> "Mesh":
        import copy
        return copy.deepcopy(self)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def center_and_scale(mesh: Mesh) -> np.ndarray:
    """obj_reader.c:239-289 in numpy: centroid to the origin, largest extent to 15 units.
    Operates in place (slot 0 included, as the reference does) and returns the 4x4 used."""
    n = mesh.num_v
    xs, ys, zs = mesh.x[1:], mesh.y[1:], mesh.z[1:]
    xc = yc = zc = 0.0
    # sequential sums, like the reference loop (numpy's pairwise sum would round differently)
    xc = float(np.add.accumulate(xs)[-1]) / n
    yc = float(np.add.accumulate(ys)[-1]) / n
    zc = float(np.add.accumulate(zs)[-1]) / n
    xd, yd, zd = xs.max() - xs.min(), ys.max() - ys.min(), zs.max() - zs.min()
    if xd > yd and xd > zd:
        delta = xd
    elif yd > zd:
        delta = yd
    else:
        delta = zd
    sf = 15 / delta
    t = np.eye(4)
    t[0, 3], t[1, 3], t[2, 3] = -xc, -yc, -zc
    s = np.diag([sf, sf, sf, 1.0])
    m = np.zeros((4, 4))
    for r in range(4):
        for c in range(4):
            acc = 0.0
            for k in range(4):
                acc = acc + s[r, k] * t[k, c]
            m[r, c] = acc
    X = m[0, 0] * mesh.x + m[0, 1] * mesh.y + m[0, 2] * mesh.z + m[0, 3]
    Y = m[1, 0] * mesh.x + m[1, 1] * mesh.y + m[1, 2] * mesh.z + m[1, 3]
    Z = m[2, 0] * mesh.x + m[2, 1] * mesh.y + m[2, 2] * mesh.z + m[2, 3]
    mesh.x, mesh.y, mesh.z = X, Y, Z
    return m


def triangle_soup(num_tris: int, seed: int = 12345, normalize: bool = True) -> Mesh:
    """Random triangle soup: centres uniform in [-7.5,7.5]^3, two edge vectors uniform in
    [-r,r]^3 with r = 15 * N^(-1/3); three private vertices and one face normal (AB x AC,
    un-normalised like obj_reader.c:51-82) per triangle."""
    rng = np.random.default_rng(seed)
    n = num_tris
    r = 15.0 * n ** (-1.0 / 3.0)
    c = rng.uniform(-7.5, 7.5, size=(n, 3))
    e1 = rng.uniform(-r, r, size=(n, 3))
    e2 = rng.uniform(-r, r, size=(n, 3))
    P = np.empty((3 * n + 1, 3))
    P[0] = 0.0
    P[1::3] = c
    P[2::3] = c + e1
    P[3::3] = c + e2
    A = np.arange(1, 3 * n + 1, 3, dtype=np.int32)
    nrm = np.cross(e1, e2)
    N = np.vstack([np.zeros((1, 3)), nrm])
    idxn = np.arange(1, n + 1, dtype=np.int32)
    m = Mesh(x=_f64(P[:, 0]), y=_f64(P[:, 1]), z=_f64(P[:, 2]),
             xn=_f64(N[:, 0]), yn=_f64(N[:, 1]), zn=_f64(N[:, 2]),
             A=_i32(A), B=_i32(A + 1), C=_i32(A + 2), An=idxn, Bn=idxn.copy(), Cn=idxn.copy(),
             name=f"soup{n}")
    if normalize:
        center_and_scale(m)
    return m


def _grid_quads(nu: int, nv: int, wrap_u: bool, wrap_v: bool):
    """Quads of a (nu x nv) parameter grid as vertex-id 4-tuples (0-based grid ids)."""
    cu = nu if wrap_u else nu - 1
    cv = nv if wrap_v else nv - 1
    i, j = np.meshgrid(np.arange(cu), np.arange(cv), indexing="ij")
    i, j = i.ravel(), j.ravel()
    i1, j1 = (i + 1) % nu, (j + 1) % nv
    return np.stack([i * nv + j, i1 * nv + j, i1 * nv + j1, i * nv + j1], axis=1)


def torus(num_tris: int, R: float = 2.0, r: float = 0.8, normalize: bool = True,
          with_uv: bool = False) -> Mesh:
    """Closed torus tessellated into quads split (a,b,c)+(a,c,d) like obj_reader.c:167-231,
    with shared vertices and smooth per-vertex normals; about `num_tris` triangles."""
    nq = max(num_tris // 2, 9)
    nu = max(int(round(np.sqrt(nq * R / r))), 3)
    nv = max(nq // nu, 3)
    th = 2 * np.pi * np.arange(nu) / nu
    ph = 2 * np.pi * np.arange(nv) / nv
    T, Pp = np.meshgrid(th, ph, indexing="ij")
    X = (R + r * np.cos(Pp)) * np.cos(T)
    Y = r * np.sin(Pp)
    Z = (R + r * np.cos(Pp)) * np.sin(T)
    NX, NY, NZ = np.cos(Pp) * np.cos(T), np.sin(Pp), np.cos(Pp) * np.sin(T)
    q = _grid_quads(nu, nv, True, True) + 1          # 1-based
    a, b, c, d = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    A = np.empty(2 * len(q), dtype=np.int32)
    B = np.empty_like(A)
    Cc = np.empty_like(A)
    A[0::2], B[0::2], Cc[0::2] = a, b, c
    A[1::2], B[1::2], Cc[1::2] = a, c, d
    pad = lambda arr: _f64(np.concatenate([[0.0], arr.ravel()]))
    m = Mesh(x=pad(X), y=pad(Y), z=pad(Z), xn=pad(NX), yn=pad(NY), zn=pad(NZ),
             A=A, B=B, C=Cc, An=A.copy(), Bn=B.copy(), Cn=Cc.copy(), name=f"torus{len(A)}")
    if with_uv:
        U, V = np.meshgrid(np.arange(nu) / nu, np.arange(nv) / nv, indexing="ij")
        m.u, m.v = pad(U), pad(V)
        m.At, m.Bt, m.Ct = A.copy(), B.copy(), Cc.copy()
    if normalize:
        center_and_scale(m)
    return m


def sphere(num_tris: int, normalize: bool = True) -> Mesh:
    """UV sphere: quads between latitude rings (split like the reference) and triangle fans
    at the poles; shared vertices, smooth normals."""
    nq = max(num_tris // 2, 8)
    nlon = max(int(round(np.sqrt(2 * nq))), 4)
    nlat = max(nq // nlon, 2)                 # number of quad rings + the two polar fans
    lat = np.pi * (np.arange(1, nlat) / nlat) - np.pi / 2
    lon = 2 * np.pi * np.arange(nlon) / nlon
    La, Lo = np.meshgrid(lat, lon, indexing="ij")
    X = np.cos(La) * np.cos(Lo)
    Y = np.sin(La)
    Z = np.cos(La) * np.sin(Lo)
    pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    south, north = len(pts) + 1, len(pts) + 2   # 1-based ids of the poles
    pts = np.vstack([np.zeros((1, 3)), pts, [[0, -1, 0]], [[0, 1, 0]]])
    tri = []
    q = _grid_quads(nlat - 1, nlon, False, True) + 1
    for a, b, c, d in q:
        tri.append((a, b, c))
        tri.append((a, c, d))
    for j in range(nlon):
        j1 = (j + 1) % nlon
        tri.append((south, 1 + j, 1 + j1))
        top = (nlat - 2) * nlon
        tri.append((north, 1 + top + j1, 1 + top + j))
    tri = np.asarray(tri, dtype=np.int32)
    m = Mesh(x=_f64(pts[:, 0]), y=_f64(pts[:, 1]), z=_f64(pts[:, 2]),
             xn=_f64(pts[:, 0]), yn=_f64(pts[:, 1]), zn=_f64(pts[:, 2]),
             A=_i32(tri[:, 0]), B=_i32(tri[:, 1]), C=_i32(tri[:, 2]),
             An=_i32(tri[:, 0]), Bn=_i32(tri[:, 1]), Cn=_i32(tri[:, 2]), name=f"sphere{len(tri)}")
    if normalize:
        center_and_scale(m)
    return m


def with_materials(mesh: Mesh, num_mats: int = 3, seed: int = 7, unassigned_every: int = 5) -> Mesh:
    """Assigns seeded constant-colour materials; every `unassigned_every`-th triangle keeps
    mtl = -1 (the 'no usemtl seen' case of obj_reader.c:18,84-90)."""
    rng = np.random.default_rng(seed)
    m = mesh.copy()
    m.Ka = rng.uniform(0.1, 1.0, size=(num_mats, 3))
    m.Kd = rng.uniform(0.1, 1.0, size=(num_mats, 3))
    m.Ks = rng.uniform(0.1, 1.0, size=(num_mats, 3))
    mt = rng.integers(0, num_mats, size=m.num_tris).astype(np.int32)
    if unassigned_every > 0:
        mt[::unassigned_every] = -1
    m.mtl = mt
    return m


def to_obj_text(mesh: Mesh, fmt: int = 3, quads: bool = False) -> str:
    """Serialises a mesh as Wavefront text in one of the four face formats of
    obj_reader.c:27-48 (1 = v, 2 = v/vt, 3 = v/vt/vn, 4 = v//vn).  With quads=True,
    consecutive triangle pairs (a,b,c),(a,c,d) are merged back into `f a b c d`."""
    out = io.StringIO()
    for i in range(1, mesh.num_v + 1):
        out.write(f"v {float(mesh.x[i])!r} {float(mesh.y[i])!r} {float(mesh.z[i])!r}\n")
    if fmt in (2, 3):
        for i in range(1, mesh.num_vt + 1):
            out.write(f"vt {float(mesh.u[i])!r} {float(mesh.v[i])!r}\n")
    if fmt in (3, 4):
        for i in range(1, mesh.num_vn + 1):
            out.write(f"vn {float(mesh.xn[i])!r} {float(mesh.yn[i])!r} {float(mesh.zn[i])!r}\n")

    def ref(k, which):
        v = (mesh.A, mesh.B, mesh.C)[which][k]
        if fmt == 1:
            return f"{v}"
        t = (mesh.At, mesh.Bt, mesh.Ct)[which][k] if mesh.At is not None else v
        n = (mesh.An, mesh.Bn, mesh.Cn)[which][k]
        return {2: f"{v}/{t}", 3: f"{v}/{t}/{n}", 4: f"{v}//{n}"}[fmt]

    k = 0
    nt = mesh.num_tris
    while k < nt:
        if (quads and k + 1 < nt and mesh.A[k] == mesh.A[k + 1] and mesh.C[k] == mesh.B[k + 1]):
            out.write(f"f {ref(k, 0)} {ref(k, 1)} {ref(k, 2)} {ref(k + 1, 2)}\n")
            k += 2
        else:
            out.write(f"f {ref(k, 0)} {ref(k, 1)} {ref(k, 2)}\n")
            k += 1
    return out.getvalue()
