"""Pin oracle/trace_oracle.c to the reference's own third-party / host code, bit for bit.

tests/golden/trace_glm_kats.npz was produced by oracle/_ref/glm_kats = g++ on the reference's vendored GLM 0.9.6.3
(/root/reference/Inference/external/include/glm) and its host source Inference/src/utilities.cpp, compiled where they
lie (oracle/Makefile `ref`, tests/golden/gen_trace_kats.py); trace_thrust_kats.npz by oracle/_ref/thrust_kats = the
image's rocThrust (same published source as the CUDA-toolkit Thrust the reference links).  These tables pin:

    a3  minstd_rand + uniform_real_distribution<float>            orc_lcg_next / orc_u01           (rng)
    a5  mat4 * vec4 of multiplyMV, glm::min/max of the box test   mulMV, glm_min/glm_max           (mulmv, minmax)
    a6  glm::intersectRayTriangle                                 glm_intersect_ray_triangle       (tri)
    a8  glm::dot/cross/length/normalize/reflect/refract           v* helpers, glm_refract          (vec)
    f1  buildTransformationMatrix, translate/rotate/scale, mat*mat, inverse, inverseTranspose      (trs, xform, matmul, inverse)

What stays citation-only (the bodies live in CUDA headers that cannot be compiled here without a stand-in
<cuda_runtime.h>): utilhash, boxIntersectionTest, sphereIntersectionTest, triangleIntersectionTest's F8 point,
calculateRandomDirectionInHemisphere, refract()/schlick()/scatterRay, the kernels of pathtrace.cu.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def L():
    return oracle._trace_lib()


@pytest.fixture(scope="module")
def glm():
    return np.load(os.path.join(GOLD, "trace_glm_kats.npz"))


def _run(L, name, x, nout):
    fn = getattr(L, "orc_kat_" + name)
    fn.argtypes = [C.c_void_p, C.c_void_p]
    fn.restype = None
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros((len(x), nout), np.float32)
    for k in range(len(x)):
        fn(x[k].ctypes.data, y[k].ctypes.data)
    return y.view(np.uint32)


def _same_bits(got, want, nan_any_payload=True):
    """Bit-equal; a NaN matches any NaN (x86 SSE and the GPU differ in the NaN payload/sign an invalid operation returns,
    and nothing downstream reads it: every comparison with a NaN is false on both sides)."""
    g, w = got.view(np.float32), want.view(np.float32)
    ok = (got == want) | (np.isnan(g) & np.isnan(w))
    return ok


@pytest.mark.parametrize("name", ["tri", "vec", "mulmv", "matmul", "trs", "xform", "inverse", "minmax"])
def test_glm_table(L, glm, name):
    x, want = glm[name + "_in"], glm[name + "_out"]
    assert len(x) >= 1000
    got = _run(L, name, x, want.shape[1])
    ok = _same_bits(got, want)
    bad = np.argwhere(~ok)
    assert ok.all(), (f"{name}: {len(bad)} mismatching words, first at row {bad[0][0]} col {bad[0][1]}: "
                      f"in {x[bad[0][0]]} got {got.view(np.float32)[tuple(bad[0])]!r} want {want.view(np.float32)[tuple(bad[0])]!r}")


def test_tri_table_has_the_interesting_cases(glm):
    o = glm["tri_out"].view(np.float32)
    assert (o[:, 0] == 1).sum() >= 1000 and (o[:, 0] == 0).sum() >= 1000


def test_thrust_rng(L):
    z = np.load(os.path.join(GOLD, "trace_thrust_kats.npz"))
    seeds, want = z["seeds"], z["out"]
    assert len(seeds) >= 1000
    got = np.zeros_like(want)
    for k, s in enumerate(seeds):
        # linear_congruential_engine::seed: s % m, 0 -> 1 (orc_seed applies it to the hash; do the same to the raw seed here)
        s0 = int(s) % 2147483647 or 1
        st = C.c_uint32(s0)
        for j in range(3):
            got[k, j] = L.orc_lcg_next(C.byref(st))
        st = C.c_uint32(s0)
        for j in range(3):
            got[k, 3 + j] = np.float32(L.orc_u01(C.byref(st), 0.0, 1.0)).view(np.uint32)
        st = C.c_uint32(s0)
        for j in range(2):
            got[k, 6 + j] = np.float32(L.orc_u01(C.byref(st), -0.5, 0.5)).view(np.uint32)
    assert np.array_equal(got, want)
    # the top states draw exactly 1.0f -- the tables hold such seeds
    assert (want[:, 3].view(np.float32) == 1.0).sum() > 0
