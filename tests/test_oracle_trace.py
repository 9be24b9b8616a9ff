"""Pin the CPU trace oracle.

The reference has no tests for this path and its headers cannot be built here without a stand-in
<cuda_runtime.h>, so the pins are the known answers SURVEY.md (Appendix B, F8) recorded from the reference
headers run in the survey container, plus the C++ standard's minstd_rand check value, plus structural
properties of pathtrace() that follow from the reference source (cited inline)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from oracle import OracleScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNELL = os.path.join(ROOT, "scenes", "cornell.txt")


@pytest.fixture(scope="module")
def L():
    return oracle._trace_lib()


def test_minstd_10000th(L):
    # thrust::minstd_rand == std::minstd_rand: 10000th draw from the default seed is 399268537
    x = C.c_uint32(1)
    for _ in range(10000):
        v = L.orc_lcg_next(C.byref(x))
    assert v == 399268537


def test_seed_and_u01_known_answers(L):
    # utilhash((1<<31)|(8<<22)|1) ^ utilhash(12345) = 939298829 -> draws 0.498972625, 0.907480478
    h = (L.orc_utilhash((1 << 31) | (8 << 22) | 1) ^ L.orc_utilhash(12345)) & 0xFFFFFFFF
    assert h == 939298829
    assert L.orc_seed(1, 12345, 8) == h % 2147483647
    s = C.c_uint32(L.orc_seed(1, 12345, 8))
    a = L.orc_u01(C.byref(s), 0.0, 1.0)
    b = L.orc_u01(C.byref(s), 0.0, 1.0)
    assert np.float32(a) == np.float32(0.498972625) and np.float32(b) == np.float32(0.907480478)
    # calculateRandomDirectionInHemisphere((0,1,0)) with the engine's next draws
    n = np.array([0, 1, 0], np.float32)
    out = np.zeros(3, np.float32)
    L.orc_hemisphere(n.ctypes.data, C.byref(s), out.ctypes.data)
    np.testing.assert_allclose(out, [-0.0120381089, 0.99536413, 0.0954217315], rtol=0, atol=2e-7)


def test_u01_top_states_round_to_one(L):
    # SURVEY 7 "minstd on GPU": float(x-1)/2^31 rounds to exactly 1.0f for the top states -- keep it.
    s = C.c_uint32(0)
    # find x with lcg(x) = m-1: x = (m-1) * inv(48271) mod m; easier: brute-force check of the mapping itself
    r = np.float32(np.float32(2147483646 - 1) / np.float32(2147483648.0))
    assert r == np.float32(1.0)


def test_det_sincos_accuracy(L):
    xs = np.linspace(0, 2 * np.pi, 20001).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    es = ec = 0.0
    for x in xs[::7]:
        L.orc_det_sincosf(C.c_float(x), C.byref(s), C.byref(c))
        es = max(es, abs(s.value - np.sin(np.float64(x))))
        ec = max(ec, abs(c.value - np.cos(np.float64(x))))
    assert es < 2.5e-7 and ec < 2.5e-7


def test_triangle_f8_quirk(L):
    # ray (0.6,0.1,1) -> -z on the unit triangle returns p=(0.1,0.3,0), t=1, correct normal weights
    f = oracle.Face()
    f.v[0][:] = [0, 0, 0]; f.v[1][:] = [1, 0, 0]; f.v[2][:] = [0, 1, 0]
    for k in range(3):
        f.n[k][:] = [0, 0, 1]
    ro = np.array([0.6, 0.1, 1.0], np.float32)
    rd = np.array([0, 0, -1], np.float32)
    P = np.zeros(3, np.float32); N = np.zeros(3, np.float32)
    t = L.orc_triangle_test(C.byref(f), ro.ctypes.data, rd.ctypes.data, P.ctypes.data, N.ctypes.data)
    assert t == pytest.approx(1.0)
    np.testing.assert_allclose(P, [0.1, 0.3, 0.0], atol=1e-6)
    np.testing.assert_allclose(N, [0, 0, 1], atol=1e-6)
    # back face is culled: glm::intersectRayTriangle rejects a < epsilon (intersect.inl:51-53)
    ro2 = np.array([0.6, 0.1, -1.0], np.float32); rd2 = np.array([0, 0, 1], np.float32)
    assert L.orc_triangle_test(C.byref(f), ro2.ctypes.data, rd2.ctypes.data, P.ctypes.data, N.ctypes.data) == -1


def test_box_and_sphere_tests(L):
    g = oracle.Geom(); g.type = oracle.CUBE
    g.translation[:] = [0, 0, 0]; g.rotation[:] = [0, 0, 0]; g.scale[:] = [2, 2, 2]
    L.orc_build_geom(C.byref(g))
    assert np.allclose(np.array(g.transform).reshape(4, 4), np.diag([2, 2, 2, 1]))
    assert np.allclose(np.array(g.inverseTransform).reshape(4, 4), np.diag([.5, .5, .5, 1]))
    ro = np.array([0, 0, 5], np.float32); rd = np.array([0, 0, -1], np.float32)
    P = np.zeros(3, np.float32); N = np.zeros(3, np.float32); o = C.c_int(0)
    t = L.orc_box_test(C.byref(g), ro.ctypes.data, rd.ctypes.data, P.ctypes.data, N.ctypes.data, C.byref(o))
    # hit the +z face at z=1 (minus getPointOnRay's 1e-4 in object space -> 2e-4 in world)
    assert t == pytest.approx(4.0, abs=1e-3) and o.value == 1
    np.testing.assert_allclose(N, [0, 0, 1], atol=1e-6)
    # from inside: exits through -z, outside flag false (intersections.h:84-88)
    ro[:] = [0, 0, 0]
    t = L.orc_box_test(C.byref(g), ro.ctypes.data, rd.ctypes.data, P.ctypes.data, N.ctypes.data, C.byref(o))
    assert t == pytest.approx(1.0, abs=1e-3) and o.value == 0
    s = oracle.Geom(); s.type = oracle.SPHERE
    s.translation[:] = [0, 0, 0]; s.rotation[:] = [0, 0, 0]; s.scale[:] = [2, 2, 2]   # radius 1
    L.orc_build_geom(C.byref(s))
    ro[:] = [0, 0, 5]
    t = L.orc_sphere_test(C.byref(s), ro.ctypes.data, rd.ctypes.data, P.ctypes.data, N.ctypes.data, C.byref(o))
    assert t == pytest.approx(4.0, abs=1e-3) and o.value == 1
    np.testing.assert_allclose(N, [0, 0, 1], atol=1e-5)
    ro[:] = [0, 0, 0]     # inside: normal flipped (intersections.h:143-145)
    t = L.orc_sphere_test(C.byref(s), ro.ctypes.data, rd.ctypes.data, P.ctypes.data, N.ctypes.data, C.byref(o))
    assert t == pytest.approx(1.0, abs=1e-3) and o.value == 0
    np.testing.assert_allclose(N, [0, 0, 1], atol=1e-5)
    ro[:] = [3, 0, 5]     # miss
    assert L.orc_sphere_test(C.byref(s), ro.ctypes.data, rd.ctypes.data, P.ctypes.data, N.ctypes.data, C.byref(o)) == -1


def test_rotated_geom_matrices(L):
    g = oracle.Geom(); g.type = oracle.CUBE
    g.translation[:] = [1, 2, 3]; g.rotation[:] = [10, 20, 30]; g.scale[:] = [.5, 2, 3]
    L.orc_build_geom(C.byref(g))
    M = np.array(g.transform, np.float64).reshape(4, 4).T      # column-major -> row/col
    Mi = np.array(g.inverseTransform, np.float64).reshape(4, 4).T
    Mit = np.array(g.invTranspose, np.float64).reshape(4, 4).T
    np.testing.assert_allclose(M @ Mi, np.eye(4), atol=1e-5)
    np.testing.assert_allclose(Mit, np.linalg.inv(M).T, atol=1e-5)
    rx, ry, rz = np.radians([10, 20, 30])
    Rx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
    Ry = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    Rz = np.array([[np.cos(rz), -np.sin(rz), 0], [np.sin(rz), np.cos(rz), 0], [0, 0, 1]])
    np.testing.assert_allclose(M[:3, :3], Rx @ Ry @ Rz @ np.diag([.5, 2, 3]), atol=1e-6)
    np.testing.assert_allclose(M[:3, 3], [1, 2, 3])


def test_cornell_camera():
    sc = OracleScene.parse(CORNELL, res=(256, 256), depth=4)
    cam = sc.camera
    # FOVY 45 used as the half-angle: yscaled = tan(45 deg) = 1 (scene.cpp:143) -> pixelLength = 2/256
    assert list(cam.pixelLength) == pytest.approx([2 / 256, 2 / 256])
    assert sc.zoom == pytest.approx(10.5) and sc.phi == pytest.approx(0.0) and sc.theta == pytest.approx(np.pi / 2)
    np.testing.assert_allclose(list(cam.position), [0, 5, 10.5], atol=1e-5)
    np.testing.assert_allclose(list(cam.view), [0, 0, -1], atol=1e-6)
    np.testing.assert_allclose(list(cam.right), [1, 0, 0], atol=1e-6)   # cross(view, (0,1,0)), un-normalised


@pytest.fixture(scope="module")
def cornell_frame():
    sc = OracleScene.parse(CORNELL, res=(128, 96), depth=4)
    return sc, sc.pathtrace()


def test_cornell_frame_structure(cornell_frame):
    sc, (g, n_live, mat0) = cornell_frame
    W, H = 128, 96
    assert g.shape == (10, H, W)
    assert n_live[0] == W * H and all(n_live[i] >= n_live[i + 1] for i in range(len(n_live) - 1))
    assert n_live[-1] == 0                      # depth exhausted: every survivor has remainingBounces 0
    hit = (mat0 >= 0).reshape(H, W)[:, ::-1]    # G-buffer is h-flipped relative to pixelIndex (pathtrace.cu:297-299)
    nlen = np.sqrt((g[3:6] ** 2).sum(0))
    assert np.allclose(nlen[hit], 1.0, atol=1e-5) and np.all(nlen[~hit] == 0)
    assert np.all(g[6][hit] > 0) and np.all(g[6][~hit] == 0)
    # albedo plane = material colour (x emittance on the light) for first hits (pathtrace.cu:379-387)
    m = mat0.reshape(H, W)[:, ::-1]
    for mid, mat in enumerate(sc.materials):
        sel = m == mid
        if not sel.any():
            continue
        if mat.emittance > 0:
            expect = np.array(mat.color[:]) * mat.emittance
        elif mat.hasRefractive or mat.hasReflective:
            continue                            # specular vs diffuse colour depends on the random branch
        else:
            expect = np.array(mat.color[:])
        for c in range(3):
            assert np.allclose(g[7 + c][sel], expect[c], atol=1e-6)
    assert np.all(g[0:3] >= 0) and g[0:3].max() <= 5.0 + 1e-5
    assert 0.05 < g[0:3].mean() < 1.0


def test_cornell_frame_deterministic_and_seeded_by_iter(cornell_frame):
    sc, (g, n_live, mat0) = cornell_frame
    g2, n2, m2 = sc.pathtrace()
    assert np.array_equal(g, g2) and np.array_equal(n_live, n2)     # F6: constant seed => identical frames
    g3, _, _ = sc.pathtrace(iter=1, depth=2)
    assert not np.array_equal(g[0:3], g3[0:3])                       # depth changes the radiance planes ...
    assert np.array_equal(g[3:10], g3[3:10])                         # ... but not the first-hit planes


def test_padded_rows_stay_zero():
    sc = OracleScene.parse(CORNELL, res=(64, 40), depth=3)
    g, n_live, _ = sc.pathtrace(pad_rows_to=64)
    assert g.shape == (10, 64, 64) and np.all(g[:, 40:, :] == 0)
    g0, _, _ = sc.pathtrace()
    assert np.array_equal(g[:, :40, :], g0)


# ------------------------------------------------------------------ the reference's optional toggles (pathtrace.cu:20-27)
def _mixed_scene(res, depth):
    from ai_path_tracer_denoiser_amd import synth
    from tests.gpu_util import add_materials
    sc = OracleScene.parse(CORNELL, res=res, depth=depth)
    first = add_materials(sc, [synth.STONE, synth.MIRROR, synth.GLASS])
    faces, lb, ub = synth.make_atrium_mesh(2048, 565, material=first, floor_material=first + 1, column_material=first + 2)
    sc.set_mesh(faces, lb, ub)
    return sc


def test_sort_material_changes_only_the_rng_order():
    sc = _mixed_scene((64, 48), 5)
    g0, n0, m0 = sc.pathtrace()
    g1, n1, m1 = sc.pathtrace(flags=oracle.TRACE_AA | oracle.TRACE_COMPACT | oracle.TRACE_SORT_MATERIAL)
    # bounce 0 is unaffected (the sort runs after it): same first hits, same survivors; later bounces draw other numbers
    assert np.array_equal(m0, m1) and n0[:2].tolist() == n1[:2].tolist()
    assert np.array_equal(g0[3:10], g1[3:10]) and not np.array_equal(g0[0:3], g1[0:3])
    assert np.isfinite(g1).all()


def test_first_bounce_cache_without_aa_changes_nothing():
    sc = _mixed_scene((48, 40), 4)
    P = 48 * 40
    cache = np.zeros(P * 36, np.uint8)
    acc_a, acc_b = np.zeros(3 * P, np.float32), np.zeros(3 * P, np.float32)
    ga, gb = np.zeros((10, 40, 48), np.float32), np.zeros((10, 40, 48), np.float32)
    for it in range(1, 4):
        sc.pathtrace(iter=it, accum=acc_a, gbuf=ga, flags=oracle.TRACE_COMPACT | oracle.TRACE_CACHE_FIRST_BOUNCE, cache=cache)
        sc.pathtrace(iter=it, accum=acc_b, gbuf=gb, flags=oracle.TRACE_COMPACT)
        assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32)), it
    assert cache.any()


def test_move_geoms():
    sc = OracleScene.parse(CORNELL)
    before = [bytes(g) for g in sc.geoms]
    sc.geoms[-1].vel[:] = [0.0, -0.1, 0.0]                    # reference scenes/Scenes/cornell.txt:123
    y0 = sc.geoms[-1].translation[1]
    sc.move_geoms(0.10)
    assert [bytes(g) for g in sc.geoms[:-1]] == before[:-1]  # zero velocity: untouched (pathtrace.cu:325-326)
    assert sc.geoms[-1].translation[1] == np.float32(np.float32(y0) + np.float32(-0.1) * np.float32(0.10))
    assert sc.geoms[-1].transform[13] == sc.geoms[-1].translation[1]


def test_reflective_material_arithmetic_is_finite_where_it_matters():
    """REFL 1 REFR 0 REFRIOR 0 (cornell_all_materials.txt:42-49): eta = inf entering, 0 leaving; the frame stays finite and
    the mirror actually reflects (paths survive the bounce)."""
    from ai_path_tracer_denoiser_amd import synth
    from tests.gpu_util import add_materials
    sc = OracleScene.parse(CORNELL, res=(64, 48), depth=6)
    m = add_materials(sc, [synth.MIRROR])
    sc.geoms[-1].materialid = m
    g, n, mat0 = sc.pathtrace()
    assert np.isfinite(g).all() and (mat0 == m).sum() > 20
    # one scatter off the mirror: the direction is the mirror reflection of the incoming one
    L = oracle._trace_lib()
    io = np.array([0, 0, 0, 0.6, -0.8, 0, 1, 1, 1], np.float32)
    hit = np.array([1.0, 0, 1, 0, 0.6, -0.8, 0], np.float32)
    rng = C.c_uint32(12345)
    L.orc_scatter(io.ctypes.data, hit.ctypes.data, C.byref(oracle.Material.from_buffer_copy(synth.MIRROR)), C.byref(rng))
    np.testing.assert_allclose(io[3:6], [0.6, 0.8, 0.0], atol=1e-6)
    np.testing.assert_allclose(io[6:9], [0.9, 0.9, 0.9], atol=0)


def test_oracle_bvh_equals_the_exhaustive_face_loop():
    """oracle/trace_bvh.c (flag TRACE_ORACLE_BVH) only accelerates the reference's loop over all faces: same G-buffer bits,
    same live counts, same first-hit materials -- including coincident faces (lowest index wins) and mixed materials."""
    from ai_path_tracer_denoiser_amd import synth
    from tests.gpu_util import add_materials
    fl = oracle.TRACE_AA | oracle.TRACE_COMPACT
    sc = OracleScene.parse(CORNELL, res=(96, 64), depth=8)
    faces, lb, ub, mats = synth.make_living_room_mesh(8192, 565, first_material=len(sc.materials))
    add_materials(sc, mats)
    dup = faces[:512].copy()
    dup["materialid"] = len(sc.materials) - 1
    sc.set_mesh(np.concatenate([faces, dup]), lb, ub)
    g0, n0, m0 = sc.pathtrace(flags=fl)
    g1, n1, m1 = sc.pathtrace(flags=fl | oracle.TRACE_ORACLE_BVH)
    assert np.array_equal(g0.view(np.uint32), g1.view(np.uint32)) and np.array_equal(n0, n1) and np.array_equal(m0, m1)
    sc2 = _mixed_scene((80, 60), 6)
    g0, n0, m0 = sc2.pathtrace(flags=fl | oracle.TRACE_SORT_MATERIAL)
    g1, n1, m1 = sc2.pathtrace(flags=fl | oracle.TRACE_SORT_MATERIAL | oracle.TRACE_ORACLE_BVH)
    assert np.array_equal(g0.view(np.uint32), g1.view(np.uint32)) and np.array_equal(n0, n1) and np.array_equal(m0, m1)


def test_dielectric_branch_known_answers():
    """The DIELECTRIC branch (interactions.h:88-168, 179-192; off in the reference, flag TRACE_DIELECTRIC here) has no golden
    vectors: its pieces are pinned to closed forms -- the unpolarised Fresnel reflectance, mirror reflection with the 0.001
    offset, Snell refraction, total internal reflection -- and the flag changes the render."""
    from ai_path_tracer_denoiser_amd import synth
    from tests.gpu_util import add_materials
    L = oracle._trace_lib()

    def fresnel(c, ei, et):
        c = float(np.clip(c, -1, 1))
        if c <= 0:
            ei, et, c = et, ei, abs(c)
        st = ei / et * np.sqrt(max(0.0, 1 - c * c))
        if st >= 1:
            return 1.0
        ct = np.sqrt(max(0.0, 1 - st * st))
        rp = (et * c - ei * ct) / (et * c + ei * ct)
        rs = (ei * c - et * ct) / (ei * c + et * ct)
        return (rp * rp + rs * rs) / 2
    for c, ei, et in [(1.0, 1.0, 1.5), (0.5, 1.0, 1.5), (0.1, 1.0, 1.33), (-0.7, 1.0, 1.5), (-0.2, 1.0, 1.5), (0.3, 1.5, 1.0), (2.0, 1.0, 1.5)]:
        assert abs(L.orc_fresnel_dielectric(c, ei, et) - fresnel(c, ei, et)) < 2e-6, (c, ei, et)
    assert abs(L.orc_fresnel_dielectric(1.0, 1.0, 1.5) - 0.04) < 1e-7           # ((n - 1) / (n + 1))^2
    assert L.orc_fresnel_dielectric(-0.2, 1.0, 1.5) == 1.0                        # leaving glass beyond the critical angle

    d = np.array([0.6, -0.8, 0.0], np.float32)
    hit = np.array([1.0, 0, 1, 0, 0.25, 0.5, 0.75], np.float32)                   # t, n = +y, P

    def scatter(mat, rng0=12345):
        io = np.concatenate([np.zeros(3, np.float32), d, np.ones(3, np.float32)]).astype(np.float32)
        rng = C.c_uint32(rng0)
        L.orc_scatter_dielectric(io.ctypes.data, hit.ctypes.data, C.byref(oracle.Material.from_buffer_copy(mat)), C.byref(rng))
        return io, rng.value
    # reflective only: mirror direction (not normalised again), origin = P + 0.001 dir, colour x specular, no random number drawn
    io, r = scatter(synth.MIRROR)
    np.testing.assert_allclose(io[3:6], [0.6, 0.8, 0.0], atol=1e-6)
    np.testing.assert_allclose(io[0:3], hit[4:7] + np.float32(.001) * io[3:6], atol=1e-7)
    np.testing.assert_allclose(io[6:9], [0.9, 0.9, 0.9], atol=0)
    assert r == 12345
    # refractive only: Snell into the denser medium
    io, r = scatter(synth.GLASS)
    eta = 1 / 1.33
    want = eta * d + (eta * 0.8 - np.sqrt(1 - eta * eta * (1 - 0.64))) * np.array([0, 1, 0])
    np.testing.assert_allclose(io[3:6], want, atol=1e-6)
    np.testing.assert_allclose(io[6:9], [0.98, 0.98, 0.98], atol=0)
    # diffuse: cosine hemisphere around the normal, colour x albedo, two random numbers
    io, r = scatter(synth.STONE)
    assert io[4] > 0 and abs(np.linalg.norm(io[3:6]) - 1) < 1e-5 and r != 12345
    np.testing.assert_allclose(io[6:9], [.75, .7, .6], atol=0)
    # both lobes: one random number picks reflection or refraction
    both = synth.material((.9, .95, 1.0), spec=(.98, .98, .98), refl=0.5, refr=1.0, ior=1.5)
    kinds = set()
    for seed in range(1, 2**31 - 2, 37000001):           # (minstd: the first draw is seed * 48271 mod (2^31 - 1))
        io, r = scatter(both, seed)
        kinds.add("reflect" if io[4] > 0 else "refract")
    assert kinds == {"reflect", "refract"}

    sc = OracleScene.parse(CORNELL, res=(64, 48), depth=6)
    first = add_materials(sc, [synth.GLASS, both])
    sc.geoms[-1].materialid = first
    sc.geoms[-2].materialid = first + 1
    fl = oracle.TRACE_AA | oracle.TRACE_COMPACT
    g0, n0, _ = sc.pathtrace(flags=fl)
    g1, n1, _ = sc.pathtrace(flags=fl | oracle.TRACE_DIELECTRIC)
    assert np.isfinite(g1).all() and not np.array_equal(g0.view(np.uint32), g1.view(np.uint32))
    g2, n2, _ = sc.pathtrace(flags=fl | oracle.TRACE_MESH_NORMAL_VIEW)
    assert np.isfinite(g2).all() and not np.array_equal(g0.view(np.uint32), g2.view(np.uint32))
    assert np.array_equal(g0[3:6].view(np.uint32), g2[3:6].view(np.uint32))     # the view changes colours only: same first-hit normals
