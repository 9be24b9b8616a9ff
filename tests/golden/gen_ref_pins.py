#!/usr/bin/env python3
"""Generate tests/golden/trace_isect_kats.npz and scene_ref_dumps.npz: known answers computed by the REFERENCE'S OWN code.

Run in the build container (needs /root/reference, g++ and the genuine NVIDIA <cuda_runtime.h> that ships inside triton's
NVIDIA backend); the fixtures it writes are data (inputs + expected outputs) and travel with the repo, the reference does not:

    make -C oracle ref            # oracle/_ref/isect_kats = g++ on the reference's unmodified Inference/src/intersections.h
                                  # oracle/_ref/scene_dump = g++ -O0 on the reference's unmodified Inference/src/scene.cpp
    python tests/golden/gen_ref_pins.py

trace_isect_kats.npz (layouts in oracle/ref_isect_kats.cpp; outputs stored as bit patterns):
    box       boxIntersectionTest       intersections.h:52-94    4096 cases: Cornell walls (SCALE .01 10 10 and permutations),
                                                                 rotations of 0/45/90, random TRS; origins outside and inside,
                                                                 grazing rays, unnormalised directions, misses
    sphere    sphereIntersectionTest    intersections.h:106-148  4096 cases, a quarter of the origins inside the sphere
    tri_full  triangleIntersectionTest  intersections.h:159-172  4096 cases: t, the F8 hit point, the interpolated normal
    aabb      RayAABBintersect          intersections.h:175-200  4096 cases incl. axis-parallel rays (1/0 = inf slabs)
    utilhash  utilhash                  intersections.h:12-20    4096 words incl. the seeds makeSeededRandomEngine forms

scene_ref_dumps.npz: for every scene file under /root/reference/Inference/scenes that the reference's own `Scene(filename)`
loads here, and for three procedural mesh scenes (OBJ text generated below), the bytes the reference's parser produced
(oracle/ref_scene_dump.cpp): Geom / Material / Face arrays, mesh_box, the loaded camera, main()'s zoom / phi / theta and the
cameras runCuda()'s orbit block derives at a few (dphi, dtheta).  The scene files themselves (data, not code) are stored beside
their dumps so that the tests can feed the same bytes to the product's and the oracle's parsers.  The reference's mesh scenes
name OBJ models that are not in its repository (scenes/Models/*.obj): those that ask for `cube.obj` are loaded with a
procedural cube placed where the scene file looks for it.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
SCENES = "/root/reference/Inference/scenes"
N = 4096
ORBITS = [(0.0, 0.0), (0.1, 0.0), (-0.35, 0.0), (0.0, 0.2), (1.3, -0.4), (3.0, 0.5)]


def run(table, x):
    x = np.ascontiguousarray(x, np.float32)
    out = subprocess.run([os.path.join(REF, "isect_kats"), table, str(len(x))], input=x.tobytes(), stdout=subprocess.PIPE,
                         check=True).stdout
    return np.frombuffer(out, np.uint32).reshape(len(x), -1).copy()


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def trs_matrix(trs):
    """T * Rx * Ry * Rz * S in double (utilities.cpp:45-52) -- only used to AIM rays, never as an expected value"""
    t, r, s = trs[0:3], np.radians(trs[3:6]), trs[6:9]
    cx, sx, cy, sy, cz, sz = np.cos(r[0]), np.sin(r[0]), np.cos(r[1]), np.sin(r[1]), np.cos(r[2]), np.sin(r[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    M = np.eye(4)
    M[:3, :3] = Rx @ Ry @ Rz @ np.diag(s)
    M[:3, 3] = t
    return M


def prim_cases(rng, n, sphere):
    t = rng.uniform(-10, 10, (n, 3))
    r = rng.uniform(-180, 180, (n, 3))
    s = np.exp(rng.uniform(np.log(0.05), np.log(12.0), (n, 3)))
    k = n // 4
    # the reference's own scenes: walls of SCALE .01 10 10 in every orientation, rotations of 0/45/90, the 3x3x3 sphere
    r[:k] = rng.choice([0.0, 45.0, 90.0, -90.0, 180.0], (k, 3))
    walls = np.array([[.01, 10, 10], [10, .01, 10], [10, 10, .01], [3, 3, 3], [3, .3, 3], [1, 1, 1], [2, 5, 2]])
    s[:k] = walls[rng.integers(0, len(walls), k)]
    t[:k] = rng.choice([0.0, 5.0, -5.0, 10.0, 2.5, -1.0], (k, 3))
    r[k:k + k // 2] = 0.0                                                          # axis-aligned, random size
    trs = np.concatenate([t, r, s], axis=1)
    o = np.zeros((n, 3)); d = np.zeros((n, 3))
    for i in range(n):
        M = trs_matrix(trs[i])
        if sphere:
            p = unit(rng, 1)[0] * 0.5 * rng.choice([1.0, 1.0, 1.0, 0.6, 0.999, 1.001, 1.3])   # on / in / just off the surface
        else:
            p = rng.uniform(-0.5, 0.5, 3) * rng.choice([1.0, 1.0, 1.0, 0.999, 1.001, 1.4])
            if rng.random() < 0.5:
                p[rng.integers(0, 3)] = rng.choice([-0.5, 0.5])                              # on a face
        target = (M @ np.append(p, 1.0))[:3]
        if rng.random() < 0.25:                                                              # origin inside the primitive
            q = (unit(rng, 1)[0] * 0.45 * rng.random()) if sphere else rng.uniform(-0.45, 0.45, 3)
            o[i] = (M @ np.append(q, 1.0))[:3]
        else:
            o[i] = target + unit(rng, 1)[0] * rng.uniform(0.05, 25.0)
        v = target - o[i]
        nv = np.linalg.norm(v)
        d[i] = v / nv if nv > 0 else unit(rng, 1)[0]
    m = n // 8
    d[:m] *= rng.uniform(0.1, 7, (m, 1))                                            # unnormalised directions
    d[m:2 * m] = unit(rng, m)                                                       # aimless: mostly misses
    ax = slice(2 * m, 2 * m + 64)
    d[ax] = np.eye(3)[rng.integers(0, 3, 64)] * rng.choice([-1.0, 1.0], (64, 1))    # axis-parallel: slabs with 1/0
    return np.concatenate([trs, o, d], axis=1).astype(np.float32)


def tri_cases(rng, n):
    v0 = rng.uniform(-5, 5, (n, 3)); e1 = rng.normal(size=(n, 3)) * rng.uniform(0.01, 3, (n, 1))
    e2 = rng.normal(size=(n, 3)) * rng.uniform(0.01, 3, (n, 1))
    v1, v2 = v0 + e1, v0 + e2
    u = rng.uniform(-0.1, 1.1, n); v = rng.uniform(-0.1, 1.1, n)
    fold = (u + v > 1) & (rng.random(n) < 0.8)
    u[fold], v[fold] = 1 - u[fold], 1 - v[fold]
    k = n // 4
    u[:k] = rng.choice([0.0, 1.0, 0.5, 1e-7, 1 - 1e-7], k); v[:k] = rng.choice([0.0, 0.5, 1e-7, 1.0], k)
    target = v0 + u[:, None] * e1 + v[:, None] * e2
    orig = target + rng.normal(size=(n, 3)) * rng.uniform(0.1, 20, (n, 1))
    d = target - orig
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[k:2 * k] *= rng.uniform(0.1, 7, (k, 1))
    back = np.einsum("ij,ij->i", e1, np.cross(d, e2)) < 0
    sw = back & (rng.random(n) < 0.8)
    v1[sw], v2[sw] = v2[sw].copy(), v1[sw].copy()
    flip = rng.random(n) < 0.1
    d[flip] *= -1
    n0, n1, n2 = unit(rng, n), unit(rng, n), unit(rng, n)
    flat = rng.random(n) < 0.3                                                      # flat-shaded: the three normals equal
    n1[flat] = n0[flat]; n2[flat] = n0[flat]
    n2[:64] *= rng.uniform(0.2, 3, (64, 1))                                         # un-normalised vertex normals
    return np.concatenate([orig, d, v0, v1, v2, n0, n1, n2], axis=1).astype(np.float32)


def aabb_cases(rng, n):
    c = rng.uniform(-5, 5, (n, 3)); h = np.exp(rng.uniform(np.log(0.01), np.log(8.0), (n, 3)))
    lb, ub = c - h, c + h
    target = c + rng.uniform(-1.3, 1.3, (n, 3)) * h
    o = target + unit(rng, n) * rng.uniform(0.05, 30, (n, 1))
    ins = rng.random(n) < 0.2
    o[ins] = (c + rng.uniform(-0.9, 0.9, (n, 3)) * h)[ins]                          # origin inside the box
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:256] = np.eye(3)[rng.integers(0, 3, 256)] * rng.choice([-1.0, 1.0], (256, 1))   # axis-parallel: dirfrac = +-inf
    d[256:512, rng.integers(0, 3)] = 0.0                                            # one zero component
    d[512:768] *= -1                                                                # box behind the ray
    # the reference's FLT_MIN upper corner (scene.cpp:216-218) on some boxes
    ub[768:832] = np.maximum(ub[768:832], np.finfo(np.float32).tiny)
    return np.concatenate([o, d, lb, ub], axis=1).astype(np.float32)


def obj_cube():
    v = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    vn = [(0, 0, -1), (0, 0, 1), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0)]
    quads = [((1, 4, 3, 2), 1), ((5, 6, 7, 8), 2), ((1, 5, 8, 4), 3), ((2, 3, 7, 6), 4), ((1, 2, 6, 5), 5), ((4, 8, 7, 3), 6)]
    s = "# procedural cube (tests/golden/gen_ref_pins.py)\n"
    s += "".join(f"v {a} {b} {c}\n" for a, b, c in v) + "".join(f"vn {a} {b} {c}\n" for a, b, c in vn)
    for q, n in quads:
        s += f"f {q[0]}//{n} {q[1]}//{n} {q[2]}//{n}\nf {q[0]}//{n} {q[2]}//{n} {q[3]}//{n}\n"
    return s


def obj_bumpy_grid(rng, nx, ny, quads=False, texcoords=False):
    """a height field with smooth-ish (un-normalised) vertex normals; optional quads (tinyobj fans them) and vt indices"""
    xs = np.linspace(-1, 1, nx + 1); ys = np.linspace(-1, 1, ny + 1)
    z = 0.2 * rng.normal(size=(ny + 1, nx + 1))
    s = "# procedural height field (tests/golden/gen_ref_pins.py)\no grid\n"
    for j in range(ny + 1):
        for i in range(nx + 1):
            s += f"v {xs[i]:.6f} {ys[j]:.6f} {z[j, i]:.6f}\n"
    for j in range(ny + 1):
        for i in range(nx + 1):
            n = np.array([-(z[j, min(i + 1, nx)] - z[j, max(i - 1, 0)]), -(z[min(j + 1, ny), i] - z[max(j - 1, 0), i]), 0.5])
            s += f"vn {n[0]:.6f} {n[1]:.6f} {n[2]:.6f}\n"                          # not unit: scene.cpp:307 normalises
    if texcoords:
        for j in range(ny + 1):
            for i in range(nx + 1):
                s += f"vt {i / nx:.4f} {j / ny:.4f}\n"
    def ix(i, j):
        k = j * (nx + 1) + i + 1
        return f"{k}/{k}/{k}" if texcoords else f"{k}//{k}"
    for j in range(ny):
        for i in range(nx):
            a, b, c, d = ix(i, j), ix(i + 1, j), ix(i + 1, j + 1), ix(i, j + 1)
            if quads and (i + j) % 3 == 0:
                s += f"f {a} {b} {c} {d}\n"
            else:
                s += f"f {a} {b} {c}\nf {a} {c} {d}\n"
    return s


MATS = ("MATERIAL 0\nRGB .98 .98 .98\nSPECEX 0\nSPECRGB 0 0 0\nREFL 0\nREFR 0\nREFRIOR 0\nEMITTANCE 5\n\n"
        "MATERIAL 1\nRGB .85 .35 .35\nSPECEX 0\nSPECRGB .98 .98 .98\nREFL 1\nREFR 0\nREFRIOR 0\nEMITTANCE 0\n\n"
        "MATERIAL 2\nRGB .35 .85 .35\nSPECEX 0\nSPECRGB .98 .98 .98\nREFL 0\nREFR 1\nREFRIOR 1.5\nEMITTANCE 0\n\n")
CAM = "CAMERA\nRES {w} {h}\nFOVY {fovy}\nITERATIONS 7\nDEPTH 6\nFILE procedural\nEYE {eye}\nLOOKAT {at}\nUP 0 1 0\n\n"


def procedural_scenes(rng):
    """(name, scene text, {relative path: file text}) -- mesh scenes with their OBJ beside the scene file"""
    out = []
    s = MATS + CAM.format(w=800, h=600, fovy=45, eye="0 5 10.5", at="0 5 0")
    s += "OBJECT 0\ncube\nmaterial 0\nTRANS 0 10 0\nROTAT 0 0 0\nSCALE 3 .3 3\n\n"
    s += "OBJECT 1\nsphere\nmaterial 2\nTRANS -1 4 -1\nROTAT 0 0 0\nSCALE 3 3 3\nVEL 0 0.5 0\n\n"
    s += "MESH 0\nPATH grid.obj\nmaterial 1\nTRANS 0.5 2 -1\nROTAT 10 20 30\nSCALE 1 2 1.5\n"
    out.append(("procedural_grid_rotated", s, {"grid.obj": obj_bumpy_grid(rng, 10, 5)}))                 # 100 faces
    s = MATS + CAM.format(w=1280, h=720, fovy=30, eye="2 3 9", at="0 2 -1")
    s += "OBJECT 0\ncube\nmaterial 0\nTRANS 0 0 0\nROTAT 0 0 90\nSCALE .01 10 10\n\n"
    s += "MESH 0\nPATH quads.obj\nmaterial 2\nTRANS -2 -3 -4\nROTAT -90 45 0\nSCALE 3 3 .5\n"
    out.append(("procedural_quads_vt_negative_quadrant", s, {"quads.obj": obj_bumpy_grid(rng, 6, 7, quads=True, texcoords=True)}))
    s = MATS + CAM.format(w=64, h=48, fovy=60, eye="0 0 6", at="0 0 0")
    s += "MESH 0\nPATH cube.obj\nmaterial 0\nTRANS 0 0 0\nROTAT 0 0 0\nSCALE 1 1 1\n"
    out.append(("procedural_cube_identity", s, {"cube.obj": obj_cube()}))
    return out


def dump_scene(workdir, scene_rel, cwd_rel="."):
    args = [os.path.join(REF, "scene_dump"), os.path.relpath(os.path.join(workdir, scene_rel), os.path.join(workdir, cwd_rel))]
    for dp, dt in ORBITS:
        args += [repr(dp), repr(dt)]
    p = subprocess.run(args, cwd=os.path.join(workdir, cwd_rel), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=60)
    return p.stdout if p.returncode == 0 and len(p.stdout) > 20 else None


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-s"])
    for exe in ("isect_kats", "scene_dump"):
        if not os.path.exists(os.path.join(REF, exe)):
            sys.exit(f"oracle/_ref/{exe} was not built (no NVIDIA cuda_runtime.h in this image?)")
    rng = np.random.default_rng(5650)
    tabs = {"box": prim_cases(rng, N, False), "sphere": prim_cases(rng, N, True), "tri_full": tri_cases(rng, N),
            "aabb": aabb_cases(rng, N)}
    words = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
    words[:16] = [0, 1, 2, 3, 0xffffffff, 0x80000000, 0x7fffffff, 12345, 1 << 15, (1 << 15) | 921599, 921599, 65535, 65536,
                  0xdeadbeef, 1 << 31, 0x165667b1]
    # makeSeededRandomEngine's argument (pathtrace.cu:52-56): (1 << 31) | (depth << 22) | iter, xor'ed with the hash of the index
    words[16:272] = [(((1 << 31) | (dpt << 22) | it) & 0xffffffff) for dpt in range(16) for it in range(1, 17)]
    tabs["utilhash"] = words.view(np.float32).reshape(-1, 1)
    out = {}
    for name, x in tabs.items():
        y = run(name, x)
        out[name + "_in"] = x.view(np.uint32) if name == "utilhash" else x
        out[name + "_out"] = y
        t = y[:, 0].view(np.float32)
        if name in ("box", "sphere"):
            hit = t != -1
            print(f"{name:9s} {x.shape} -> {y.shape}  hits {int(hit.sum())}  from inside {int((hit & (y[:, 7].view(np.float32) == 0)).sum())}"
                  f"  nan rows {int(np.isnan(y.view(np.float32)).any(axis=1).sum())}")
        elif name == "tri_full":
            print(f"{name:9s} {x.shape} -> {y.shape}  hits {int((t != -1).sum())}")
        elif name == "aabb":
            print(f"{name:9s} {x.shape} -> {y.shape}  hits {int((t == 1).sum())}")
    np.savez_compressed(os.path.join(HERE, "trace_isect_kats.npz"), **out)

    # ---- scenes
    scenes = {}
    with tempfile.TemporaryDirectory() as tmp:
        # the reference's own scene files, laid out as its repository lays them out; cube.obj where its mesh scenes look for it
        os.makedirs(os.path.join(tmp, "scenes", "Models")); os.makedirs(os.path.join(tmp, "build"))
        cube = obj_cube()
        with open(os.path.join(tmp, "scenes", "Models", "cube.obj"), "w") as f:
            f.write(cube)
        for sub in sorted(os.listdir(SCENES)):
            d = os.path.join(SCENES, sub)
            if not os.path.isdir(d):
                continue
            os.makedirs(os.path.join(tmp, "scenes", sub), exist_ok=True)
            for fn in sorted(os.listdir(d)):
                src = os.path.join(d, fn)
                if not os.path.isfile(src):
                    continue
                data = open(src, "rb").read()
                rel = os.path.join("scenes", sub, fn)
                with open(os.path.join(tmp, rel), "wb") as f:
                    f.write(data)
                dump = dump_scene(tmp, rel, "build")           # cwd = build/: "../scenes/Models/cube.obj" resolves
                if dump is None:
                    print(f"  reference parser does not load {sub}/{fn} here (missing OBJ model or malformed file)")
                    continue
                uses_cube = b"cube.obj" in data
                scenes[f"{sub}/{fn}"] = (data, {"scenes/Models/cube.obj": cube} if uses_cube else {}, rel, "build", dump)
        for name, text, files in procedural_scenes(rng):
            d = os.path.join(tmp, name); os.makedirs(d)
            with open(os.path.join(d, "scene.txt"), "w") as f:
                f.write(text)
            for rel, body in files.items():
                with open(os.path.join(d, rel), "w") as f:
                    f.write(body)
            dump = dump_scene(d, "scene.txt", ".")
            assert dump is not None, name
            scenes[name] = (text.encode(), files, "scene.txt", ".", dump)
    out = {"names": np.array(sorted(scenes))}
    for k, name in enumerate(sorted(scenes)):
        data, files, rel, cwd, dump = scenes[name]
        out[f"s{k}_text"] = np.frombuffer(data, np.uint8)
        out[f"s{k}_rel"] = np.array([rel, cwd])
        out[f"s{k}_aux_names"] = np.array(sorted(files)) if files else np.array([], dtype="U1")
        for j, fn in enumerate(sorted(files)):
            out[f"s{k}_aux{j}"] = np.frombuffer(files[fn].encode(), np.uint8)
        out[f"s{k}_dump"] = np.frombuffer(dump, np.uint8)
        ng, nm, nf = np.frombuffer(dump[:12], np.int32)
        print(f"  {name:45s} geoms {ng:3d} materials {nm:3d} faces {nf:4d}  dump {len(dump)} B")
    np.savez_compressed(os.path.join(HERE, "scene_ref_dumps.npz"), **out)
    print("scenes:", len(scenes))


if __name__ == "__main__":
    sys.exit(main())
