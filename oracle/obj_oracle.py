"""Pure-Python restatement of the reference's Wavefront loader (obj_reader.c), for small
files.  TEST INFRASTRUCTURE ONLY -- the product's loader is raytracer_amd/csrc/rt_obj.cpp.

obj_reader.c cannot be compiled here (MSVC CRT calls and SDL headers), so this follows its
source text: the scanf-driven second pass (obj_reader.c:628-726), scan_face (:93-237) and
center_and_scale_object (:239-289).  The one behaviour pinned by a recorded run of the
reference's scan_face is the quad split: `f 1 2 3 4` -> (1,2,3) and (1,3,4).
Format-1 faces store the vertex index, not the address the reference stores (:98-100).
"""
from __future__ import annotations

import os
import re

import numpy as np

_INT = re.compile(rb"[ \t\r\n\f\v]*([+-]?\d+)")
_REAL = re.compile(rb"[ \t\r\n\f\v]*([+-]?(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?))")
_TOK = re.compile(rb"[ \t\r\n\f\v]*([^ \t\r\n\f\v]+)")


class _Scan:
    """The handful of fscanf conversions the loader uses."""

    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def token(self):                      # "%s"
        m = _TOK.match(self.d, self.p)
        if not m:
            return None
        self.p = m.end()
        return m.group(1).decode("latin-1")

    def integer(self):                    # "%d"
        m = _INT.match(self.d, self.p)
        if not m:
            return None
        self.p = m.end()
        return int(m.group(1))

    def real(self):                       # "%lf"
        m = _REAL.match(self.d, self.p)
        if not m:
            return None
        self.p = m.end()
        return float(m.group(1))

    def lit(self, ch: bytes):             # a literal character of the format string
        if self.d[self.p:self.p + 1] == ch:
            self.p += 1
            return True
        return False

    def skip_line(self):                  # fgets
        nl = self.d.find(b"\n", self.p)
        self.p = len(self.d) if nl < 0 else nl + 1


def _face_format(s: _Scan) -> int:        # get_face_format, :27-48
    save = s.p
    try:
        if s.integer() is None:
            return 0
        n = 1
        if s.lit(b"/"):
            if s.integer() is not None:
                n = 2
                if s.lit(b"/") and s.integer() is not None:
                    n = 3
        if n == 1:
            return 4 if s.lit(b"/") else 1
        return n
    finally:
        s.p = save


def _vertex(s: _Scan, fmt: int):
    save = s.p
    v = s.integer()
    vt = vn = 0
    ok = v is not None
    if ok and fmt == 2:
        ok = s.lit(b"/") and (vt := s.integer()) is not None
    elif ok and fmt == 3:
        ok = (s.lit(b"/") and (vt := s.integer()) is not None and s.lit(b"/")
              and (vn := s.integer()) is not None)
    elif ok and fmt == 4:
        ok = s.lit(b"/") and s.lit(b"/") and (vn := s.integer()) is not None
    if not ok:
        s.p = save
        return None
    return v, vt, vn


class ObjError(Exception):
    pass


def load(path: str) -> dict:
    data = open(path, "rb").read()
    base = os.path.dirname(path)
    s = _Scan(data)
    x, y, z = [0.0], [0.0], [0.0]
    xn, yn, zn = [0.0], [0.0], [0.0]
    u, v = [0.0], [0.0]
    tris = []                      # (A,B,C, At,Bt,Ct, An,Bn,Cn, mtl)
    mats = []                      # dict(name, Ka, Kd, Ks)
    current = -1                   # :18
    while (key := s.token()) is not None:
        if key == "v":
            p = [s.real() for _ in range(3)]
            s.real()               # optional w, :669
            x.append(p[0] or 0.0); y.append(p[1] or 0.0); z.append(p[2] or 0.0)
        elif key == "vt":
            a, b = s.real(), s.real()
            u.append(a or 0.0); v.append(b or 0.0)
        elif key == "vn":
            p = [s.real() for _ in range(3)]
            xn.append(p[0] or 0.0); yn.append(p[1] or 0.0); zn.append(p[2] or 0.0)
        elif key in ("g", "s"):
            s.token()
        elif key == "f":
            fmt = _face_format(s)
            vs = []
            while len(vs) < 5 and (r := _vertex(s, fmt)) is not None:
                vs.append(r)
            if len(vs) > 4:
                raise ObjError("The object contains a face with more than 4 vertices")   # :500-505
            if len(vs) < 3:
                raise ObjError("face with fewer than 3 vertices")
            if fmt in (1, 2):      # find_and_add_normal, :51-82: AB x AC from the raw coordinates
                A, B, Cc = vs[0][0], vs[1][0], vs[2][0]
                AB = (x[B] - x[A], y[B] - y[A], z[B] - z[A])
                AC = (x[Cc] - x[A], y[Cc] - y[A], z[Cc] - z[A])
                xn.append(AB[1] * AC[2] - AC[1] * AB[2])
                yn.append(AC[0] * AB[2] - AB[0] * AC[2])
                zn.append(AB[0] * AC[1] - AC[0] * AB[1])
                vs = [(a, b, len(xn) - 1) for a, b, _ in vs]
            order = [(0, 1, 2)] + ([(0, 2, 3)] if len(vs) == 4 else [])      # :167-231
            for i, j, k in order:
                tris.append((vs[i][0], vs[j][0], vs[k][0], vs[i][1], vs[j][1], vs[k][1],
                             vs[i][2], vs[j][2], vs[k][2], current))
        elif key == "mtllib":
            name = s.token()
            mats += _read_mtl(os.path.join(base, name))
        elif key == "usemtl":
            name = s.token()
            for i, m in enumerate(mats):   # :710-715
                if m["name"] == name:
                    current = i
                    break
        elif key.startswith("#"):
            s.skip_line()
    X, Y, Z = (np.array(a, dtype=np.float64) for a in (x, y, z))
    n = len(X) - 1
    # center_and_scale_object, :239-289 (sequential sums; slot 0 is transformed too)
    xc = yc = zc = 0.0
    for i in range(1, n + 1):
        xc += X[i]; yc += Y[i]; zc += Z[i]
    xc /= n; yc /= n; zc /= n
    xd, yd, zd = X[1:].max() - X[1:].min(), Y[1:].max() - Y[1:].min(), Z[1:].max() - Z[1:].min()
    delta = xd if (xd > yd and xd > zd) else (yd if yd > zd else zd)
    sf = 15 / delta
    t = np.eye(4); t[0, 3], t[1, 3], t[2, 3] = -xc, -yc, -zc
    sm = np.diag([sf, sf, sf, 1.0])
    m = np.zeros((4, 4))
    for r in range(4):
        for c in range(4):
            acc = 0.0
            for k in range(4):
                acc = acc + sm[r, k] * t[k, c]
            m[r, c] = acc
    Xo = m[0, 0] * X + m[0, 1] * Y + m[0, 2] * Z + m[0, 3]
    Yo = m[1, 0] * X + m[1, 1] * Y + m[1, 2] * Z + m[1, 3]
    Zo = m[2, 0] * X + m[2, 1] * Y + m[2, 2] * Z + m[2, 3]
    T = np.array(tris, dtype=np.int32).reshape(-1, 10)
    return {
        "x": Xo, "y": Yo, "z": Zo,
        "xn": np.array(xn), "yn": np.array(yn), "zn": np.array(zn),
        "u": np.array(u), "v": np.array(v),
        "A": T[:, 0], "B": T[:, 1], "C": T[:, 2], "At": T[:, 3], "Bt": T[:, 4], "Ct": T[:, 5],
        "An": T[:, 6], "Bn": T[:, 7], "Cn": T[:, 8], "mtl": T[:, 9],
        "Ka": np.array([m_["Ka"] for m_ in mats]).reshape(-1, 3),
        "Kd": np.array([m_["Kd"] for m_ in mats]).reshape(-1, 3),
        "Ks": np.array([m_["Ks"] for m_ in mats]).reshape(-1, 3),
        "mat_names": [m_["name"] for m_ in mats],
    }


def _read_mtl(path: str) -> list:          # read_mtl_file, :383-453
    s = _Scan(open(path, "rb").read())
    mats = []
    while (key := s.token()) is not None:
        if key == "newmtl":
            mats.append({"name": s.token(), "Ka": [1.0] * 3, "Kd": [1.0] * 3, "Ks": [1.0] * 3})   # :291-303
        elif key in ("Ka", "Kd", "Ks") and mats:
            vals = [s.real() for _ in range(3)]
            for i, val in enumerate(vals):
                if val is not None:
                    mats[-1][key][i] = val
        elif key in ("Ke", "Tf"):
            [s.real() for _ in range(3)]
        elif key in ("map_Ka", "map_Kd", "map_Ks"):
            s.token()              # image decoding is not restated (SDL_image, :352-381)
        elif key in ("Tr", "Ns", "Ni", "d"):
            s.real()
        elif key == "illum":
            s.integer()
        else:
            s.skip_line()
    return mats
