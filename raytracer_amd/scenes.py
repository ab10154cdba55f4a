"""Seeded synthetic scenes for the BASELINE configs whose assets are not in the
reference tree (sponza/dragon/buddha are listed in .MISSING_LARGE_BLOBS) and that
cannot be read on the GPU box anyway.  Pure numpy; the output goes through the
host layer (Model.from_arrays -> BVHBuilder) like any loaded OBJ.

  atrium()        C3/C5 "Sponza-class" interior, ~262 k triangles, 2 materials,
                  normals + texture coordinates
  triangle_soup() C4 random triangle soup, no normals / tex coords
  subdivide()     midpoint subdivision (C2: bunny x2 -> 80 016 triangles)
  checker_texture() / noise_texture()  RGB8 albedo stand-ins for F16s.bmp / uv_debug.png
"""
import math

import numpy as np


class _Mesh:
    def __init__(self):
        self.pos, self.nrm, self.uv, self.tri, self.mat = [], [], [], [], []
        self.nv = 0

    def add_grid(self, P, N, UV, mat, flip=False):
        """P, N: (nu+1, nv+1, 3); UV: (nu+1, nv+1, 2). Two triangles per cell."""
        nu, nv = P.shape[0] - 1, P.shape[1] - 1
        idx = (np.arange((nu + 1) * (nv + 1)).reshape(nu + 1, nv + 1) + self.nv).astype(np.uint32)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        t1 = np.stack([a, b, c], -1).reshape(-1, 3)
        t2 = np.stack([a, c, d], -1).reshape(-1, 3)
        t = np.concatenate([t1, t2])
        if flip:
            t = t[:, ::-1]
        self.tri.append(t)
        self.mat.append(np.full(len(t), mat, np.uint32))
        self.pos.append(P.reshape(-1, 3)); self.nrm.append(N.reshape(-1, 3)); self.uv.append(UV.reshape(-1, 2))
        self.nv += (nu + 1) * (nv + 1)

    def arrays(self):
        tri = np.concatenate(self.tri).astype(np.uint32)
        mat = np.concatenate(self.mat)
        triangles = np.concatenate([tri, mat[:, None]], 1).astype(np.uint32)
        pos = np.concatenate(self.pos).astype(np.float32)
        nrm = np.concatenate(self.nrm).astype(np.float32)
        nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20).astype(np.float32)
        uv = np.concatenate(self.uv).astype(np.float32)
        return triangles, pos, uv, nrm.astype(np.float32)


def _uvgrid(nu, nv):
    u, v = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="ij")
    return u, v


def _plane(m, origin, eu, ev, nu, nv, mat, uvscale=(1.0, 1.0)):
    u, v = _uvgrid(nu, nv)
    origin, eu, ev = (np.asarray(x, np.float64) for x in (origin, eu, ev))
    P = origin + u[..., None] * eu + v[..., None] * ev
    n = np.cross(eu, ev)
    N = np.broadcast_to(n / np.linalg.norm(n), P.shape)
    UV = np.stack([u * uvscale[0], v * uvscale[1]], -1)
    m.add_grid(P, N, UV, mat)


def _cylinder(m, base, radius, height, seg, rings, mat, flute=0.0):
    u, v = _uvgrid(seg, rings)
    ang = 2 * np.pi * u
    r = radius * (1.0 + flute * np.cos(16 * ang)) * (1.0 - 0.08 * v)
    P = np.stack([base[0] + r * np.cos(ang), base[1] + height * v, base[2] + r * np.sin(ang)], -1)
    N = np.stack([np.cos(ang), np.full_like(ang, 0.08 * radius / height), np.sin(ang)], -1)
    m.add_grid(P, N, np.stack([u * 2, v * 4], -1), mat, flip=True)


def _arch(m, c0, c1, y, rad_tube, seg, tube, mat):
    """Half-torus spanning from c0 to c1 (points on the xz plane) rising above height y."""
    c0, c1 = np.asarray(c0, np.float64), np.asarray(c1, np.float64)
    mid = (c0 + c1) / 2
    axis = (c1 - c0)
    R = np.linalg.norm(axis) / 2
    axis /= (2 * R)
    u, v = _uvgrid(seg, tube)
    th = np.pi * u          # along the arch
    ph = 2 * np.pi * v      # around the tube
    # centre line
    cx = mid[0] - axis[0] * R * np.cos(th)
    cz = mid[1] - axis[1] * R * np.cos(th)
    cy = y + R * np.sin(th)
    # local frame: radial (in the arch plane) and side (perpendicular, horizontal)
    rad = np.stack([-axis[0] * np.cos(th), np.sin(th), -axis[1] * np.cos(th)], -1)
    side = np.array([-axis[1], 0.0, axis[0]])
    Nn = rad * np.cos(ph)[..., None] + side * np.sin(ph)[..., None]
    P = np.stack([cx, cy, cz], -1) + rad_tube * Nn
    m.add_grid(P, Nn, np.stack([u * 3, v], -1), mat)


def _cloth(m, rng, origin, width, drop, n, mat):
    u, v = _uvgrid(n, n)
    ph = rng.uniform(0, 2 * np.pi, 4)
    amp = rng.uniform(0.15, 0.35, 2)
    z = amp[0] * np.sin(6 * np.pi * u + ph[0]) * (0.3 + v) + amp[1] * np.sin(9 * np.pi * v + ph[1] + 3 * u) * 0.5
    P = np.stack([origin[0] + width * u, origin[1] - drop * v, origin[2] + z], -1)
    du = np.gradient(P, axis=0); dv = np.gradient(P, axis=1)
    N = np.cross(du, dv)
    m.add_grid(P, N, np.stack([u, v], -1), mat)


def _sphere(m, centre, radius, seg, rings, mat):
    u, v = _uvgrid(seg, rings)
    ang, pol = 2 * np.pi * u, np.pi * (v * 0.998 + 0.001)
    N = np.stack([np.cos(ang) * np.sin(pol), np.cos(pol), np.sin(ang) * np.sin(pol)], -1)
    P = np.asarray(centre, np.float64) + radius * N
    m.add_grid(P, N, np.stack([u, v], -1), mat, flip=True)


def atrium(seed=20240):
    """Procedural two-storey colonnaded hall.  Returns (triangles[n,4] u32, positions,
    tex_coords, normals, camera dict).  262 944 triangles for the default seed."""
    rng = np.random.default_rng(seed)
    m = _Mesh()
    L, Wd, Hh = 20.0, 8.0, 12.0
    _plane(m, (-L, 0, -Wd), (0, 0, 2 * Wd), (2 * L, 0, 0), 64, 128, 0, (8, 20))            # floor
    # roofs over the two aisles only: the nave is open to the sky (like Sponza's
    # courtyard), so sun light and sun shadows reach the floor
    for z0, z1 in ((-Wd, -4.0), (4.0, Wd)):
        _plane(m, (-L, Hh, z0), (2 * L, 0, 0), (0, 0, z1 - z0), 128, 16, 0, (10, 1))
    # side walls out of long thin strips (bad-for-BVH primitives, as in real interiors)
    for z, sgn in ((-Wd, 1.0), (Wd, -1.0)):
        for k in range(200):
            y0, y1 = Hh * k / 200.0, Hh * (k + 1) / 200.0
            if sgn > 0:
                _plane(m, (-L, y0, z), (2 * L, 0, 0), (0, y1 - y0, 0), 1, 1, 0, (20, 0.05))
            else:
                _plane(m, (-L, y0, z), (0, y1 - y0, 0), (2 * L, 0, 0), 1, 1, 0, (0.05, 20))
    # colonnade: 2 rows x 10 columns x 2 storeys, arches between neighbours
    xs = np.linspace(-L + 2.0, L - 2.0, 10)
    for z in (-4.5, 4.5):
        for storey in range(2):
            y0 = storey * 6.0
            for x in xs:
                _cylinder(m, (x, y0, z), 0.45, 4.0, 48, 32, 1, flute=0.04)
            for x0, x1 in zip(xs[:-1], xs[1:]):
                _arch(m, (x0, z), (x1, z), y0 + 4.0, 0.22, 32, 16, 1)
        # balcony floor of the upper storey
        zlo, zhi = (z - 3.5, z + 1.0) if z < 0 else (z - 1.0, z + 3.5)
        _plane(m, (-L, 5.95, zlo), (0, 0, zhi - zlo), (2 * L, 0, 0), 16, 64, 0, (2, 20))
    # hanging cloth
    for k in range(6):
        x = -15.0 + 5.5 * k
        _cloth(m, rng, (x, 10.5, rng.uniform(-2.5, 2.5)), 4.0, 5.0 + rng.uniform(-1, 1), 64, 1)
    # statues
    for k in range(4):
        _sphere(m, (-12.0 + 8.0 * k, 1.0, rng.uniform(-1.5, 1.5)), 1.0, 48, 64, 1 if k % 2 else 0)
    triangles, pos, uv, nrm = m.arrays()
    camera = {"position": (-18.0, 3.0, 0.4), "direction": (0.985, 0.02, -0.05), "fov_deg": 90.0}
    return triangles, pos, uv, nrm, camera


def triangle_soup(n, seed=1, extent=100.0, size=0.5):
    """n triangles: centre ~U([-extent,extent]^3) + 3 offsets ~U([-size,size]^3); no shared vertices."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(-extent, extent, (n, 1, 3))
    o = rng.uniform(-size, size, (n, 3, 3))
    pos = (c + o).reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    triangles = np.concatenate([idx, np.zeros((n, 1), np.uint32)], 1)
    camera = {"position": (0.0, 0.0, -2.5 * extent), "direction": (0.0, 0.0, 1.0), "fov_deg": 60.0}
    return triangles, pos, camera


def subdivide(triangles, positions, normals=None, levels=1):
    """Midpoint subdivision: every triangle -> 4; normals re-interpolated and normalised."""
    tri = np.asarray(triangles, np.uint32).reshape(-1, 4)
    pos = np.asarray(positions, np.float32).reshape(-1, 3)
    nrm = None if normals is None else np.asarray(normals, np.float32).reshape(-1, 3)
    for _ in range(levels):
        e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]).astype(np.int64)
        e.sort(axis=1)
        key = e[:, 0] * (len(pos) + 1) + e[:, 1]
        uniq, inv = np.unique(key, return_inverse=True)
        ue = np.stack([uniq // (len(pos) + 1), uniq % (len(pos) + 1)], 1)
        mid = ((pos[ue[:, 0]] + pos[ue[:, 1]]) * np.float32(0.5)).astype(np.float32)
        base = len(pos)
        n = len(tri)
        m01, m12, m20 = (inv[:n] + base).astype(np.uint32), (inv[n:2 * n] + base).astype(np.uint32), (inv[2 * n:] + base).astype(np.uint32)
        if nrm is not None:
            nm = nrm[ue[:, 0]] + nrm[ue[:, 1]]
            nm = nm / np.maximum(np.linalg.norm(nm, axis=1, keepdims=True), 1e-20)
            nrm = np.concatenate([nrm, nm.astype(np.float32)])
        pos = np.concatenate([pos, mid])
        a, b, c, mt = tri[:, 0], tri[:, 1], tri[:, 2], tri[:, 3]
        tri = np.concatenate([np.stack([a, m01, m20, mt], 1), np.stack([m01, b, m12, mt], 1),
                              np.stack([m20, m12, c, mt], 1), np.stack([m01, m12, m20, mt], 1)]).astype(np.uint32)
    return tri, pos, nrm


def checker_texture(size=1024, cells=16, seed=7):
    rng = np.random.default_rng(seed)
    pal = rng.integers(60, 256, (cells, cells, 3), dtype=np.int64)
    y, x = np.mgrid[0:size, 0:size]
    t = pal[(y * cells) // size, (x * cells) // size]
    shade = 0.85 + 0.15 * np.sin(x * 0.05) * np.cos(y * 0.07)
    return np.clip(t * shade[..., None], 0, 255).astype(np.uint8)


def noise_texture(width=512, height=256, seed=11):
    rng = np.random.default_rng(seed)
    base = rng.integers(96, 256, (height // 32, width // 32, 3)).astype(np.float64)
    t = np.kron(base, np.ones((32, 32, 1)))
    t = 0.85 * t + 0.15 * rng.integers(0, 256, (height, width, 3))
    return np.clip(t, 0, 255).astype(np.uint8)


def framing_camera(positions, dist, offset=(1.0, 0.4, 0.6), fov_deg=120.0):
    """The camera the golden fixtures were made with (tests/golden/make_golden.py:camera_for, same arithmetic): the
    reference's defaults — fovy 120 degrees, up +y (src/rendering/camera.rs:6-7,34-42) — placed `dist` bounding-box
    radii from the model's centre along `offset`, looking at the centre.  Returns (position f32[3], direction f32[3], fov)."""
    pos = np.asarray(positions).reshape(-1, 3)
    lo, hi = pos.min(0), pos.max(0)
    c = (lo + hi) / 2
    r = np.linalg.norm(hi - lo) / 2
    off = np.asarray(offset, np.float64)
    eye = c + dist * r * off / np.linalg.norm(off)
    d = c - eye
    d /= np.linalg.norm(d)
    return eye.astype(np.float32), d.astype(np.float32), math.radians(fov_deg)
