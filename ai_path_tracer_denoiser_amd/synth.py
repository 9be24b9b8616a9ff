"""Deterministic synthetic weights and G-buffer inputs.

The reference ships no trained weights and no dataset, so tests,
golden-vector generation and the bench all draw from this generator.  Only integer
hashing and exactly-rounded fp32 operations (mul/add/div/sqrt) are used, so every
machine regenerates bit-identical arrays; nothing large needs committing.

Weight statistics follow the reference's init (training/train.py:32-38): conv weights
with Kaiming fan-in variance 2/fan_in (uniform here instead of normal so that no libm
call is involved), bias 0.01.  BatchNorm parameters are *randomised* (the reference
init is gamma=1, beta=0, mean=0, var=1) so that parity tests exercise every term,
including negative gamma.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import arch

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x):
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30)
    x *= _M1
    x ^= x >> np.uint64(27)
    x *= _M2
    x ^= x >> np.uint64(31)
    return x


def uniform01(n, seed, stream):
    """n fp32 values in [0,1), a pure function of (seed, stream, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _mix64(np.array([np.uint64(seed) * _GOLD + np.uint64(stream)], dtype=np.uint64))[0]
        h = _mix64(idx * _GOLD + key)
    return ((h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def make_params(seed=565, bn_random=True):
    """Synthetic parameters for the 28 (conv, BN) pairs, keyed by arch.layer_table() names."""
    params = OrderedDict()
    for li, (name, _, _, cin, cout) in enumerate(arch.layer_table()):
        fan_in = cin * 9
        bound = np.float32(np.sqrt(np.float32(6.0) / np.float32(fan_in)))   # uniform with var 2/fan_in
        u = uniform01(cout * cin * 9, seed, 16 * li + 0)
        w = ((u - np.float32(0.5)) * np.float32(2.0) * bound).astype(np.float32).reshape(cout, cin, 3, 3)
        b = np.full(cout, 0.01, np.float32)
        if bn_random:
            g = np.float32(0.5) + uniform01(cout, seed, 16 * li + 1)            # 0.5 .. 1.5
            sgn = np.where(uniform01(cout, seed, 16 * li + 2) < np.float32(0.1), np.float32(-1), np.float32(1))
            gamma = (g * sgn).astype(np.float32)
            beta = ((uniform01(cout, seed, 16 * li + 3) - np.float32(0.5)) * np.float32(0.4)).astype(np.float32)
            mean = ((uniform01(cout, seed, 16 * li + 4) - np.float32(0.5)) * np.float32(0.2)).astype(np.float32)
            var = (np.float32(0.5) + uniform01(cout, seed, 16 * li + 5)).astype(np.float32)
        else:
            gamma = np.ones(cout, np.float32)
            beta = np.zeros(cout, np.float32)
            mean = np.zeros(cout, np.float32)
            var = np.ones(cout, np.float32)
        params[name] = dict(w=w, b=b, gamma=gamma, beta=beta, mean=mean, var=var)
    return params


def make_blob(seed=565, bn_random=True) -> bytes:
    return arch.pack_blob(make_params(seed, bn_random))


_PALETTE = np.array([[.98, .98, .98], [.85, .35, .35], [.35, .85, .35], [5., 5., 5.], [.75, .7, .6]],
                    dtype=np.float32)   # Cornell materials (scenes/Scenes/cornell.txt) + light + Sponza stone


def make_gbuffer(H, W, seed=0, frame=0):
    """A G-buffer-like [10,H,W] fp32 input: noisy RGB in [0,1] with 2% light
    pixels at 5.0, unit normals, depth in [0,25), palette albedo, 10% all-zero miss pixels."""
    n = H * W
    s = 1000 + 97 * frame
    g = np.zeros((10, n), np.float32)
    for c in range(3):
        g[c] = uniform01(n, seed, s + c)
    light = uniform01(n, seed, s + 3) < np.float32(0.02)
    g[0:3, light] = np.float32(5.0)
    v = np.stack([uniform01(n, seed, s + 4 + c) * np.float32(2) - np.float32(1) for c in range(3)])
    nrm = np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]).astype(np.float32)
    nrm = np.where(nrm == 0, np.float32(1), nrm)
    g[3:6] = (v / nrm).astype(np.float32)
    g[6] = uniform01(n, seed, s + 7) * np.float32(25.0)
    pal = (uniform01(n, seed, s + 8) * np.float32(len(_PALETTE))).astype(np.int64)
    pal = np.minimum(pal, len(_PALETTE) - 1)
    g[7:10] = _PALETTE[pal].T
    miss = uniform01(n, seed, s + 9) < np.float32(0.10)
    g[:, miss] = 0
    return g.reshape(10, H, W)


# --------------------------------------------------------------------------------------------- procedural meshes
# The reference's mesh scenes point at OBJ files that do not ship (Sponza / living room exist only as GIFs),
# so the mesh configurations use seeded procedural stand-ins with a stated triangle count.

FACE_DTYPE = np.dtype([("v", np.float32, (3, 3)), ("n", np.float32, (3, 3)), ("materialid", np.int32)])   # Face, 76 B
assert FACE_DTYPE.itemsize == 76


def _grid_surface(fn, nu, nv, material, flip=False):
    """Tessellate a parametric surface (u,v in [0,1]) -> (position, normal) into 2*nu*nv smooth-shaded triangles,
    counter-clockwise seen from the side the normal points to (the reference culls back faces, intersect.inl:51-53)."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    pos, nrm = fn(uu, vv)                                   # [nu+1, nv+1, 3]
    nrm = nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    i, j = i.ravel(), j.ravel()
    quads = [(i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)]
    tris = [(0, 1, 2), (0, 2, 3)] if not flip else [(0, 2, 1), (0, 3, 2)]
    faces = np.zeros(2 * nu * nv, FACE_DTYPE)
    for t, tri in enumerate(tris):
        for k, corner in enumerate(tri):
            a, b = quads[corner]
            faces["v"][t::2, k] = pos[a, b]
            faces["n"][t::2, k] = nrm[a, b]
    faces["materialid"] = material
    # fix the winding so the geometric normal agrees with the shading normal
    e1 = faces["v"][:, 1] - faces["v"][:, 0]
    e2 = faces["v"][:, 2] - faces["v"][:, 0]
    gn = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    bad = (gn * faces["n"][:, 0].astype(np.float64)).sum(-1) < 0
    faces["v"][bad] = faces["v"][bad][:, [0, 2, 1]]
    faces["n"][bad] = faces["n"][bad][:, [0, 2, 1]]
    return faces


def make_atrium_mesh(n_triangles=262144, seed=565, material=1, floor_material=None, column_material=None):
    """"Sponza-like" atrium inside the 10 x 10 x 10 Cornell volume: two rows of columns, arches between neighbouring
    columns, a gently rippled floor slab and a few spheres.  Returns (faces[FACE_DTYPE], lb[3], ub[3]) with exactly
    n_triangles triangles (the floor grid absorbs the remainder).  Deterministic for a given (n_triangles, seed).
    floor_material / column_material (default: material) give the floor slab and the columns their own material ids --
    BASELINE configs[3] ("reflective Sponza") sets both to the reflective material."""
    assert n_triangles >= 2048
    floor_material = material if floor_material is None else floor_material
    column_material = material if column_material is None else column_material
    rng_u = uniform01(64, seed, 9001).astype(np.float64)
    parts = []
    ncol = 6
    budget = n_triangles
    col_tris = int(budget * 0.45 / (2 * ncol))
    arch_tris = int(budget * 0.30 / (2 * (ncol - 1)))
    sph_tris = int(budget * 0.10 / 3)

    def dims(target):                                           # nu, nv with 2*nu*nv <= target, nu ~ 2*nv
        nv = max(2, int(np.sqrt(target / 4.0)))
        nu = max(3, target // (2 * nv))
        return nu, nv

    xs = np.linspace(-3.6, 3.6, ncol)
    for row, z in enumerate((-2.6, 1.4)):
        for ci, x in enumerate(xs):
            r = 0.28 + 0.06 * rng_u[row * ncol + ci]
            nu, nv = dims(col_tris)

            def col(u, v, x=x, z=z, r=r):
                ang = 2 * np.pi * u
                bulge = 1.0 + 0.08 * np.sin(np.pi * v) + 0.03 * np.cos(8 * ang)
                p = np.stack([x + r * bulge * np.cos(ang), 0.02 + 6.0 * v, z + r * bulge * np.sin(ang)], -1)
                n = np.stack([np.cos(ang), -0.08 * np.pi * np.cos(np.pi * v) * r / 6.0, np.sin(ang)], -1)
                return p, n
            parts.append(_grid_surface(col, nu, nv, column_material))
        for ci in range(ncol - 1):
            x0, x1 = xs[ci], xs[ci + 1]
            nu, nv = dims(arch_tris)

            def arch(u, v, x0=x0, x1=x1, z=z):
                th = np.pi * u                                  # along the arch
                ph = 2 * np.pi * v                              # around the tube
                R, rt = 0.5 * (x1 - x0), 0.16
                cx, cy = 0.5 * (x0 + x1), 6.0
                dirx, diry = -np.cos(th), np.sin(th)            # centre-line direction from the arch centre
                p = np.stack([cx + (R + rt * np.cos(ph)) * dirx, cy + (R + rt * np.cos(ph)) * diry,
                              z + rt * np.sin(ph)], -1)
                n = np.stack([np.cos(ph) * dirx, np.cos(ph) * diry, np.sin(ph)], -1)
                return p, n
            parts.append(_grid_surface(arch, nu, nv, material))
    for si in range(3):
        cx, cz = -2.5 + 2.5 * si, -0.6 + 0.5 * rng_u[40 + si]
        rad = 0.55 + 0.2 * rng_u[44 + si]
        nu, nv = dims(sph_tris)

        def sph(u, v, cx=cx, cz=cz, rad=rad):
            ang, pol = 2 * np.pi * u, np.pi * (0.02 + 0.96 * v)
            n = np.stack([np.sin(pol) * np.cos(ang), np.cos(pol), np.sin(pol) * np.sin(ang)], -1)
            return np.array([cx, rad + 0.05, cz]) + rad * n, n
        parts.append(_grid_surface(sph, nu, nv, material))
    used = sum(len(p) for p in parts)
    rest = n_triangles - used
    assert rest >= 2 and rest % 2 == 0, (n_triangles, used)
    # floor slab: nu x nv grid with 2*nu*nv == rest exactly (factor rest/2)
    half = rest // 2
    nu = int(np.sqrt(half))
    while half % nu:
        nu -= 1
    nv = half // nu

    def floor(u, v):
        x, z = -4.6 + 9.2 * u, -4.6 + 9.2 * v
        y = 0.03 + 0.015 * np.sin(3.0 * x) * np.cos(2.5 * z)
        dydx = 0.015 * 3.0 * np.cos(3.0 * x) * np.cos(2.5 * z)
        dydz = -0.015 * 2.5 * np.sin(3.0 * x) * np.sin(2.5 * z)
        return np.stack([x, y, z], -1), np.stack([-dydx, np.ones_like(x), -dydz], -1)
    parts.append(_grid_surface(floor, nu, nv, floor_material))
    faces = np.concatenate(parts)
    assert len(faces) == n_triangles
    verts = faces["v"].reshape(-1, 3)
    lb = verts.min(axis=0)
    # Scene::loadObj starts the upper bound at FLT_MIN, the smallest positive float (scene.cpp:216-218)
    ub = np.maximum(verts.max(axis=0), np.float32(np.finfo(np.float32).tiny))
    return faces, lb.astype(np.float32), ub.astype(np.float32)


def _mesh_bounds(faces):
    verts = faces["v"].reshape(-1, 3)
    lb = verts.min(axis=0)
    # Scene::loadObj starts the upper bound at FLT_MIN, the smallest positive float (scene.cpp:216-218)
    ub = np.maximum(verts.max(axis=0), np.float32(np.finfo(np.float32).tiny))
    return lb.astype(np.float32), ub.astype(np.float32)


# Material records in the reference's 44-byte layout (sceneStructs.h:46-56); bytes(...) of one element converts to the
# ctypes Material of api.py / oracle via from_buffer_copy.
MATERIAL_DTYPE = np.dtype([("color", np.float32, 3), ("specex", np.float32), ("speccolor", np.float32, 3),
                           ("refl", np.float32), ("refr", np.float32), ("ior", np.float32), ("emit", np.float32)])
assert MATERIAL_DTYPE.itemsize == 44


def material(rgb, spec=(0, 0, 0), refl=0.0, refr=0.0, ior=0.0, emit=0.0) -> bytes:
    m = np.zeros(1, MATERIAL_DTYPE)
    m["color"] = rgb; m["speccolor"] = spec; m["refl"] = refl; m["refr"] = refr; m["ior"] = ior; m["emit"] = emit
    return m.tobytes()


STONE = material((.75, .7, .6))                                          # all-diffuse stone
# `REFL 1 SPECRGB .9 .9 .9` with REFR 0 REFRIOR 0 as the reference's reflective material
# (scenes/Scenes/cornell_all_materials.txt:42-49): scatterRay then runs refract() with eta = 1/0 = inf or 0
MIRROR = material((.98, .98, .98), spec=(.9, .9, .9), refl=1.0)
GLASS = material((.9, .95, 1.0), spec=(.98, .98, .98), refr=1.0, ior=1.33)   # `REFR 1 REFRIOR 1.33`
FABRIC = material((.55, .25, .2))
WOOD = material((.6, .45, .3))
WHITE = material((.9, .9, .85))


def make_living_room_mesh(n_triangles=524288, seed=565, first_material=0):
    """"Living-room-like" interior inside the 10 x 10 x 10 Cornell volume (BASELINE configs[4]): mixed diffuse,
    reflective and refractive face materials.  Returns (faces, lb, ub, materials) where materials is the list of 44-byte
    material records the mesh uses; face material ids start at first_material (append the list to the scene's materials).
    Parts: wooden floor, rippled rug, three sofa cushions and a back rest (fabric), a glass table top on four chrome legs,
    two glass vases (surfaces of revolution), a wall mirror, a chrome torus ornament, a white lamp shade.
    Deterministic for a given (n_triangles, seed); the rug grid absorbs the remainder so the count is exact."""
    assert n_triangles >= 4096
    mats = [WOOD, FABRIC, GLASS, MIRROR, WHITE, STONE]
    WOOD_I, FABRIC_I, GLASS_I, MIRROR_I, WHITE_I, STONE_I = [first_material + k for k in range(6)]
    r = uniform01(64, seed, 9101).astype(np.float64)
    parts = []

    def dims(target, aspect=2.0):
        nv = max(2, int(np.sqrt(target / (2.0 * aspect))))
        nu = max(3, int(target // (2 * nv)))
        return nu, nv

    def ellipsoid(c, a):
        c, a = np.asarray(c, float), np.asarray(a, float)

        def f(u, v):
            ang, pol = 2 * np.pi * u, np.pi * (0.01 + 0.98 * v)
            d = np.stack([np.sin(pol) * np.cos(ang), np.cos(pol), np.sin(pol) * np.sin(ang)], -1)
            return c + a * d, d / a
        return f

    def revolve(c, profile, dprofile, height):
        c = np.asarray(c, float)

        def f(u, v):
            ang = 2 * np.pi * u
            rad, drad = profile(v), dprofile(v)
            p = np.stack([c[0] + rad * np.cos(ang), c[1] + height * v, c[2] + rad * np.sin(ang)], -1)
            n = np.stack([np.cos(ang), -drad / height, np.sin(ang)], -1)
            return p, n
        return f

    def plane_xy(x0, x1, y0, y1, z):
        def f(u, v):
            p = np.stack([x0 + (x1 - x0) * u, y0 + (y1 - y0) * v, np.full_like(u, z)], -1)
            return p, np.broadcast_to(np.array([0.0, 0.0, 1.0]), p.shape).copy()
        return f

    def torus(c, R, rt):
        c = np.asarray(c, float)

        def f(u, v):
            th, ph = 2 * np.pi * u, 2 * np.pi * v
            n = np.stack([np.cos(ph) * np.cos(th), np.sin(ph), np.cos(ph) * np.sin(th)], -1)
            p = c + np.stack([(R + rt * np.cos(ph)) * np.cos(th), rt * np.sin(ph), (R + rt * np.cos(ph)) * np.sin(th)], -1)
            return p, n
        return f

    B = n_triangles

    def add(fn, share, mat, aspect=2.0):
        nu, nv = dims(int(B * share), aspect)
        parts.append(_grid_surface(fn, nu, nv, mat))

    for k in range(3):                                                            # sofa cushions
        add(ellipsoid([-2.2 + 2.2 * k, 1.0 + 0.05 * r[k], -2.6], [1.0, 0.45, 0.9]), 0.07, FABRIC_I)
    add(ellipsoid([0.0, 2.0, -3.5], [3.3, 1.0, 0.35]), 0.07, FABRIC_I)            # back rest
    add(ellipsoid([0.3, 1.55, 0.6], [1.7, 0.07, 1.0]), 0.08, GLASS_I)             # glass table top
    for k, (lx, lz) in enumerate([(-0.9, 0.0), (1.5, 0.0), (-0.9, 1.2), (1.5, 1.2)]):   # chrome legs
        add(revolve([lx, 0.03, lz], lambda v: 0.06 + 0.0 * v, lambda v: 0.0 * v, 1.45), 0.015, MIRROR_I)
    for k, (vx, vz) in enumerate([(3.2, -1.0), (-3.4, 1.8)]):                     # glass vases
        h = 1.6 + 0.4 * r[10 + k]
        add(revolve([vx, 0.03, vz], lambda v: 0.25 + 0.2 * np.sin(np.pi * v) ** 2 + 0.05 * np.cos(3 * np.pi * v),
                    lambda v: 0.2 * np.pi * np.sin(2 * np.pi * v) - 0.15 * np.pi * np.sin(3 * np.pi * v), h), 0.07, GLASS_I)
    add(plane_xy(-2.5, 2.5, 3.2, 6.2, -4.55), 0.04, MIRROR_I, aspect=1.0)         # wall mirror
    add(torus([0.3, 1.85, 0.6], 0.45, 0.12), 0.08, MIRROR_I)                      # chrome ornament on the table
    add(ellipsoid([3.4, 5.2, -3.2], [0.7, 0.9, 0.7]), 0.06, WHITE_I)              # lamp shade
    add(ellipsoid([-3.4, 0.6, -0.4], [0.6, 0.6, 0.6]), 0.04, STONE_I)             # pouffe

    def floor(u, v):
        p = np.stack([-4.7 + 9.4 * u, np.full_like(u, 0.02), -4.7 + 9.4 * v], -1)
        return p, np.broadcast_to(np.array([0.0, 1.0, 0.0]), p.shape).copy()
    add(floor, 0.10, WOOD_I, aspect=1.0)
    used = sum(len(p) for p in parts)
    rest = n_triangles - used
    assert rest >= 2 and rest % 2 == 0, (n_triangles, used)
    half = rest // 2
    nu = int(np.sqrt(half))
    while half % nu:
        nu -= 1
    nv = half // nu

    def rug(u, v):
        x, z = -2.8 + 6.0 * u, -1.2 + 4.2 * v
        y = 0.06 + 0.012 * np.sin(9.0 * x) * np.sin(7.0 * z)
        dydx = 0.012 * 9.0 * np.cos(9.0 * x) * np.sin(7.0 * z)
        dydz = 0.012 * 7.0 * np.sin(9.0 * x) * np.cos(7.0 * z)
        return np.stack([x, y, z], -1), np.stack([-dydx, np.ones_like(x), -dydz], -1)
    parts.append(_grid_surface(rug, nu, nv, FABRIC_I))
    faces = np.concatenate(parts)
    assert len(faces) == n_triangles
    lb, ub = _mesh_bounds(faces)
    return faces, lb, ub, mats


def write_obj(path, faces):
    """Write faces as a Wavefront OBJ (v / vn / f a//a) for the scene-file front end (MESH blocks)."""
    with open(path, "w") as f:
        for fa in faces:
            for k in range(3):
                f.write("v %.9g %.9g %.9g\n" % tuple(fa["v"][k]))
        for fa in faces:
            for k in range(3):
                f.write("vn %.9g %.9g %.9g\n" % tuple(fa["n"][k]))
        for i in range(len(faces)):
            a = 3 * i + 1
            f.write(f"f {a}//{a} {a + 1}//{a + 1} {a + 2}//{a + 2}\n")
