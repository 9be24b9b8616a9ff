#!/usr/bin/env python3
"""Generate tests/golden/trace_glm_kats.npz and trace_thrust_kats.npz: known answers computed by the REFERENCE'S OWN code.

Run in the build container (needs /root/reference and hipcc); the fixtures it writes are data (inputs + expected
outputs) and travel with the repo, the reference does not:

    make -C oracle ref            # oracle/_ref/glm_kats  = g++ on the reference's vendored GLM + Inference/src/utilities.cpp
                                  # oracle/_ref/thrust_kats = hipcc --offload-host-only on the image's rocThrust
    python tests/golden/gen_trace_kats.py

Tables (2048 seeded cases each, tri 3072, rng 4166; layouts in oracle/ref_glm_kats.cpp):
    tri      glm::intersectRayTriangle                     rays aimed at/around random triangles, grazing, back-facing,
                                                           parallel, degenerate, unnormalised directions
    vec      dot, cross, length, normalize, reflect, refract  unit and non-unit vectors; eta in {0, inf, 1/1.5, 1.5, 1/1.33, 1.33, random}
                                                           (REFRIOR 0 of the reference's reflective materials gives eta = inf or 0)
    mulmv    mat4 * vec4 (multiplyMV)                      TRS matrices and their inverses, w in {0, 1}
    matmul   mat4 * mat4
    trs      utilityCore::buildTransformationMatrix + glm::inverse + glm::inverseTranspose   (scene.cpp:92-95)
    xform    glm::translate / rotate / scale on a general matrix
    inverse  glm::inverse / glm::inverseTranspose of general matrices
    minmax   glm::min / glm::max incl. NaN, +-0, inf
    rng      thrust::default_random_engine + uniform_real_distribution<float>: raw outputs and U(0,1), U(-.5,.5) draws
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
N = 2048


def run_glm(table, x):
    x = np.ascontiguousarray(x, np.float32)
    out = subprocess.run([os.path.join(REF, "glm_kats"), table, str(len(x))], input=x.tobytes(), stdout=subprocess.PIPE,
                         check=True).stdout
    return np.frombuffer(out, np.float32).reshape(len(x), -1).copy()


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def trs_inputs(rng, n):
    t = rng.uniform(-10, 10, (n, 3))
    r = rng.uniform(-180, 180, (n, 3))
    s = np.exp(rng.uniform(np.log(0.01), np.log(12.0), (n, 3)))
    # the reference's own scenes: axis-aligned, rotations of 0/45/90, walls of scale .01 x 10 x 10 (cornell.txt)
    k = n // 4
    r[:k] = rng.choice([0.0, 45.0, 90.0, -90.0, 180.0], (k, 3))
    s[:k] = rng.choice([0.01, 1.0, 3.0, 10.0], (k, 3))
    t[:k] = rng.choice([0.0, 5.0, -5.0, 10.0, 2.5], (k, 3))
    return np.concatenate([t, r, s], axis=1).astype(np.float32)


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-s"])
    rng = np.random.default_rng(565)
    tabs = {}

    # ---- tri
    NT = 3072
    v0 = rng.uniform(-5, 5, (NT, 3)); e1 = rng.normal(size=(NT, 3)) * rng.uniform(0.01, 3, (NT, 1))
    e2 = rng.normal(size=(NT, 3)) * rng.uniform(0.01, 3, (NT, 1))
    v1, v2 = v0 + e1, v0 + e2
    u = rng.uniform(-0.1, 1.1, NT); v = rng.uniform(-0.1, 1.1, NT)
    fold = (u + v > 1) & (rng.random(NT) < 0.8)                                  # most targets inside the triangle
    u[fold], v[fold] = 1 - u[fold], 1 - v[fold]
    k = NT // 4
    u[:k] = rng.choice([0.0, 1.0, 0.5, 1e-7, 1 - 1e-7], k); v[:k] = rng.choice([0.0, 0.5, 1e-7, 1.0], k)   # edges and corners
    target = v0 + u[:, None] * e1 + v[:, None] * e2
    orig = target + rng.normal(size=(NT, 3)) * rng.uniform(0.1, 20, (NT, 1))
    d = target - orig
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[k:2 * k] *= rng.uniform(0.1, 7, (k, 1))                                   # unnormalised directions
    back = np.einsum("ij,ij->i", e1, np.cross(d, e2)) < 0                       # back-facing: culled (a < epsilon)
    sw = back & (rng.random(NT) < 0.8)                                           # turn most of them front-facing
    v1[sw], v2[sw] = v2[sw].copy(), v1[sw].copy()
    e1, e2 = v1 - v0, v2 - v0
    flip = rng.random(NT) < 0.15
    d[flip] *= -1                                                               # pointing away (t < 0)
    par = slice(2 * k, 2 * k + 64)
    d[par] = e1[par] / np.linalg.norm(e1[par], axis=1, keepdims=True)           # parallel to the plane
    deg = slice(2 * k + 64, 2 * k + 128)
    v2[deg] = v1[deg]                                                           # degenerate triangles
    tabs["tri"] = np.concatenate([orig, d, v0, v1, v2], axis=1).astype(np.float32)

    # ---- vec
    a = unit(rng, N); b = unit(rng, N)
    a[N // 2:] *= rng.uniform(0.01, 30, (N - N // 2, 1)).astype(np.float32)     # non-unit first operand (normalize/length)
    b[3 * N // 4:] *= rng.uniform(0.5, 2, (N - 3 * N // 4, 1)).astype(np.float32)
    a[:32] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 32)] * rng.choice([-1.0, 1.0], (32, 1))
    b[:32] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 32)]
    a[32:40] = 0.0                                                              # normalize(0) -> NaN, as GLM does
    b[40:72] = -a[40:72]                                                        # head-on: 1 - d*d = 0 (inf * 0 in refract)
    eta = rng.choice([0.0, np.inf, 1 / 1.5, 1.5, 1 / 1.33, 1.33, 1.0], N).astype(np.float32)
    eta[N // 2:] = rng.uniform(0.2, 3.0, N - N // 2)
    tabs["vec"] = np.concatenate([a, b, eta[:, None]], axis=1).astype(np.float32)

    # ---- trs (first: its outputs feed mulmv / matmul / inverse)
    tabs["trs"] = trs_inputs(rng, N)
    trs_out = run_glm("trs", tabs["trs"])
    mats = np.concatenate([trs_out[:, :16], trs_out[:, 16:32], trs_out[:, 32:48]], axis=0)      # 3N matrices
    pick = mats[rng.integers(0, len(mats), N)]
    vec = np.concatenate([rng.uniform(-10, 10, (N, 3)), rng.choice([0.0, 1.0], (N, 1))], axis=1)
    tabs["mulmv"] = np.concatenate([pick, vec], axis=1).astype(np.float32)
    tabs["matmul"] = np.concatenate([mats[rng.integers(0, len(mats), N)], mats[rng.integers(0, len(mats), N)]], axis=1)
    gen = mats[rng.integers(0, len(mats), N)].copy()
    gen[N // 2:] = rng.normal(size=(N - N // 2, 16))                             # general (non-affine) matrices too
    tabs["inverse"] = gen.astype(np.float32)
    ang = rng.uniform(-np.pi, np.pi, (N, 1)); ang[:64] = rng.choice([0.0, np.pi / 2, np.pi / 4, np.pi], (64, 1))
    axis = rng.normal(size=(N, 3)); axis[:96] = np.eye(3)[rng.integers(0, 3, 96)]
    tabs["xform"] = np.concatenate([gen, ang, axis], axis=1).astype(np.float32)

    # ---- minmax
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 1e38, -1e38, 1e-45], np.float32)
    g = np.stack(np.meshgrid(sp, sp), -1).reshape(-1, 2)
    tabs["minmax"] = np.concatenate([g, rng.normal(size=(N - len(g), 2)).astype(np.float32)]).astype(np.float32)

    out = {}
    for name, x in tabs.items():
        y = trs_out if name == "trs" else run_glm(name, x)
        out[name + "_in"] = x
        out[name + "_out"] = y.view(np.uint32)                   # bit patterns: NaNs compare equal as data
        print(f"{name:8s} {x.shape} -> {y.shape}  nan rows {int(np.isnan(y).any(axis=1).sum())}")
    print("tri hits:", int(out["tri_out"].view(np.float32)[:, 0].sum()), "of", len(out["tri_in"]))
    np.savez_compressed(os.path.join(HERE, "trace_glm_kats.npz"), **out)

    # ---- rng: seeds as makeSeededRandomEngine produces them (any uint32) + the edge seeds of the LCG
    seeds = rng.integers(0, 2**32, 4096, dtype=np.uint64).astype(np.uint32)
    seeds[:8] = [0, 1, 2147483646, 2147483647, 2147483648, 4294967295, 12345, 939298829]
    raw = subprocess.run([os.path.join(REF, "thrust_kats"), str(len(seeds))], input=seeds.tobytes(),
                         stdout=subprocess.PIPE, check=True).stdout
    o = np.frombuffer(raw, np.uint32).reshape(len(seeds), 8).copy()
    # the states whose U(0,1) draw rounds to exactly 1.0f (SURVEY 7 "minstd on GPU"): seed s such that the first output is
    # in the top ~64 states; found by inverting the LCG: s = x * inv(48271) mod m
    m = 2147483647
    inv = pow(48271, -1, m)
    top = np.array([(x * inv) % m for x in range(m - 70, m)], np.uint32)
    raw = subprocess.run([os.path.join(REF, "thrust_kats"), str(len(top))], input=top.tobytes(),
                         stdout=subprocess.PIPE, check=True).stdout
    o2 = np.frombuffer(raw, np.uint32).reshape(len(top), 8).copy()
    seeds = np.concatenate([seeds, top]); o = np.concatenate([o, o2])
    print("rng", o.shape, "draws == 1.0f:", int((o[:, 3].view(np.float32) == 1.0).sum()))
    np.savez_compressed(os.path.join(HERE, "trace_thrust_kats.npz"), seeds=seeds, out=o)


if __name__ == "__main__":
    sys.exit(main())
