"""-m gpu: the material branches and the optional toggles of the reference's hot loop, bit-exact against the CPU oracle.

* reflective material exactly as the reference writes it (`REFL 1 REFR 0 REFRIOR 0`, cornell_all_materials.txt:42-49):
  scatterRay runs refract() with eta = 1/0 = inf (entering) or 0 (leaving), i.e. inf*0 = NaN and sqrt(-inf) arithmetic
  (interactions.h:74-85,116-120,208-236) -- BASELINE configs[3] (reflective Sponza) and a Cornell-only variant;
* mixed diffuse / reflective / refractive mesh faces -- BASELINE configs[4] (living room), a <= 16k-face subset bit-exact
  against the oracle's brute-force loop, the 524 288-face scene through BVH == GPU brute force and properties;
* SORT_MATERIAL, CACHE_BOUNCE, MOTION_BLUR (pathtrace.cu:20-27) as run-time flags.
"""
import numpy as np
import pytest

from ai_path_tracer_denoiser_amd import api, synth
from tests.gpu_util import CORNELL, add_materials, bits, gpu_trace

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _same(g, g_ref, what=""):
    a, b = bits(g), bits(g_ref)
    assert np.array_equal(a, b), f"{what}: {np.count_nonzero(a != b)} words differ"


def _cornell(res, depth):
    import oracle
    return oracle.OracleScene.parse(CORNELL, res=res, depth=depth)


def test_reflective_sphere_cornell(ctx):
    sc = _cornell((160, 120), 8)
    m = add_materials(sc, [synth.MIRROR])
    assert sc.geoms[-1].type == 0                       # the sphere
    sc.geoms[-1].materialid = m
    g_ref, n_ref, m_ref = sc.pathtrace()
    g, n, mt = gpu_trace(ctx, sc, 8)
    assert np.array_equal(mt, m_ref) and n[:len(n_ref)].tolist() == n_ref.tolist()
    _same(g, g_ref, "reflective sphere")
    assert (m_ref == m).sum() > 150


def test_reflective_walls_and_floor_cornell(ctx):
    """every wall a mirror: long specular chains, rays that hit surfaces from behind (eta = 0 branch)"""
    sc = _cornell((128, 96), 8)
    m = add_materials(sc, [synth.MIRROR])
    for g in sc.geoms[1:6]:
        g.materialid = m
    g_ref, n_ref, m_ref = sc.pathtrace()
    g, n, mt = gpu_trace(ctx, sc, 8)
    assert n[:len(n_ref)].tolist() == n_ref.tolist()
    _same(g, g_ref, "mirror room")


def test_reflective_sponza_like_mesh_configs3(ctx):
    """BASELINE configs[3] shape at oracle size: atrium with reflective floor and columns."""
    sc = _cornell((128, 96), 8)
    first = add_materials(sc, [synth.STONE, synth.MIRROR])
    faces, lb, ub = synth.make_atrium_mesh(8192, 565, material=first, floor_material=first + 1, column_material=first + 1)
    sc.set_mesh(faces, lb, ub)
    g_ref, n_ref, m_ref = sc.pathtrace()
    g, n, mt = gpu_trace(ctx, sc, 8)
    assert np.array_equal(mt, m_ref) and n[:len(n_ref)].tolist() == n_ref.tolist()
    _same(g, g_ref, "reflective atrium")
    assert (m_ref == first + 1).sum() > 1000            # the reflective faces are visible


def test_reflective_sponza_full_size_bvh_equals_brute_force_sample(ctx):
    """configs[3] at its real size: 262 144 triangles.  The oracle's O(F) loop is out of reach at 1280x720, so the full frame is
    checked for determinism and finiteness, and a 160x90 frame of the SAME mesh against the product's brute-force loop."""
    sc = _cornell((160, 90), 8)
    first = add_materials(sc, [synth.STONE, synth.MIRROR])
    faces, lb, ub = synth.make_atrium_mesh(262144, 565, material=first, floor_material=first + 1, column_material=first + 1)
    sc.set_mesh(faces, lb, ub)
    flags = api.TRACE_DEFAULT | api.TRACE_RECORD_MAT0
    g1, n1, m1 = gpu_trace(ctx, sc, 8, flags=flags)
    g2, n2, m2 = gpu_trace(ctx, sc, 8, flags=flags | api.TRACE_BRUTE_FORCE, upload=False)
    assert np.array_equal(n1, n2) and np.array_equal(m1, m2)
    _same(g1, g2, "BVH vs brute force")
    sc.camera.res[:] = [1280, 720]
    import oracle
    import ctypes as C
    oracle._trace_lib().orc_camera_setup(C.byref(sc.camera), C.c_float(sc.fovy))
    sc.set_orbit(sc.zoom, sc.phi, sc.theta)
    g3, n3, m3 = gpu_trace(ctx, sc, 8)
    g4, n4, _ = gpu_trace(ctx, sc, 8, upload=False)
    _same(g3, g4, "repeat")
    assert np.isfinite(g3).all() and n3[0] == 1280 * 720 and (m3 == first + 1).sum() > 100000


def _living_room(res, depth, ntri):
    sc = _cornell(res, depth)
    faces, lb, ub, mats = synth.make_living_room_mesh(ntri, 565, first_material=len(sc.materials))
    add_materials(sc, mats)
    sc.set_mesh(faces, lb, ub)
    return sc


def test_living_room_subset_configs4(ctx):
    """BASELINE configs[4] materials (diffuse + reflective + refractive faces) on a 16 384-face living room, depth 12."""
    sc = _living_room((128, 72), 12, 16384)
    g_ref, n_ref, m_ref = sc.pathtrace()
    g, n, mt = gpu_trace(ctx, sc, 12)
    assert np.array_equal(mt, m_ref) and n[:len(n_ref)].tolist() == n_ref.tolist()
    _same(g, g_ref, "living room")
    seen = set(np.unique(m_ref).tolist())
    assert {7, 8} <= seen                                # glass and mirror faces are first hits somewhere


def test_living_room_full_size_1920x1080_depth12(ctx):
    """configs[4] as named: 524 288 faces, 1920x1080, depth 12.  BVH == GPU brute force on a 160x90 frame of the same mesh;
    the full frame is deterministic, finite, fully accounted for (n_live) and sees every material class."""
    sc = _living_room((160, 90), 12, 524288)
    flags = api.TRACE_DEFAULT | api.TRACE_RECORD_MAT0
    g1, n1, m1 = gpu_trace(ctx, sc, 12, flags=flags)
    g2, n2, m2 = gpu_trace(ctx, sc, 12, flags=flags | api.TRACE_BRUTE_FORCE, upload=False)
    assert np.array_equal(n1, n2) and np.array_equal(m1, m2)
    _same(g1, g2, "BVH vs brute force")
    import oracle
    import ctypes as C
    sc.camera.res[:] = [1920, 1080]
    oracle._trace_lib().orc_camera_setup(C.byref(sc.camera), C.c_float(sc.fovy))
    sc.set_orbit(sc.zoom, sc.phi, sc.theta)
    g3, n3, m3 = gpu_trace(ctx, sc, 12)
    g4, n4, _ = gpu_trace(ctx, sc, 12, upload=False)
    _same(g3, g4, "repeat")
    assert np.isfinite(g3).all() and n3[0] == 1920 * 1080
    assert all(n3[k] >= n3[k + 1] for k in range(12))
    assert {7, 8} <= set(np.unique(m3).tolist())


# ----------------------------------------------------------------------------------------------- toggles (pathtrace.cu:20-27)
def _mixed_scene(res, depth):
    sc = _cornell(res, depth)
    first = add_materials(sc, [synth.STONE, synth.MIRROR, synth.GLASS])
    faces, lb, ub = synth.make_atrium_mesh(4096, 565, material=first, floor_material=first + 1, column_material=first + 2)
    sc.set_mesh(faces, lb, ub)
    return sc


def test_sort_material_matches_the_reference_order(ctx):
    """SORT_MATERIAL: the RNG index of a path is its slot after thrust::partition + the stable sort_by_key on the
    pre-partition hit records (pathtrace.cu:351,505-510) -- reproduced by the oracle and by the product's counting sort."""
    import oracle
    sc = _mixed_scene((160, 96), 6)
    fl = oracle.TRACE_AA | oracle.TRACE_COMPACT | oracle.TRACE_SORT_MATERIAL
    g_ref, n_ref, m_ref = sc.pathtrace(flags=fl)
    g, n, mt = gpu_trace(ctx, sc, 6, flags=api.TRACE_DEFAULT | api.TRACE_SORT_MATERIAL | api.TRACE_RECORD_MAT0)
    assert np.array_equal(mt, m_ref) and n[:len(n_ref)].tolist() == n_ref.tolist()
    _same(g, g_ref, "sorted")
    g0, _, _ = sc.pathtrace()
    assert not np.array_equal(g0[0:3], g_ref[0:3]) and np.array_equal(g0[3:10], g_ref[3:10])   # the order matters from bounce 1 on
    with pytest.raises(api.AiptError):
        gpu_trace(ctx, sc, 6, flags=api.TRACE_AA | api.TRACE_SORT_MATERIAL, upload=False)        # needs COMPACT


def test_first_bounce_cache(ctx):
    """CACHE_BOUNCE: iterations 2.. reuse the bounce-0 hit records of iteration 1; legal only without AA (assert, :435)."""
    import oracle
    sc = _mixed_scene((128, 80), 5)
    W, H = 128, 80
    fl_o = oracle.TRACE_COMPACT | oracle.TRACE_CACHE_FIRST_BOUNCE
    cache = np.zeros(W * H * 36, np.uint8)
    accum = np.zeros(3 * W * H, np.float32)
    g_ref = np.zeros((10, H, W), np.float32)
    refs = []
    for it in range(1, 4):
        sc.pathtrace(iter=it, accum=accum, gbuf=g_ref, flags=fl_o, cache=cache)
        refs.append(g_ref.copy())
    got = []
    gpu_trace(ctx, sc, 5, flags=api.TRACE_COMPACT | api.TRACE_CACHE_FIRST_BOUNCE, iters=3,
              each_iter=lambda it, g: got.append(g.copy()))
    for it in range(3):
        _same(got[it], refs[it], f"cached, iteration {it + 1}")
    plain = []
    gpu_trace(ctx, sc, 5, flags=api.TRACE_COMPACT, iters=3, each_iter=lambda it, g: plain.append(g.copy()))
    _same(plain[2], got[2], "cache on vs off")          # without AA the primary rays repeat, so the cache changes nothing
    with pytest.raises(api.AiptError):
        gpu_trace(ctx, sc, 5, flags=api.TRACE_DEFAULT | api.TRACE_CACHE_FIRST_BOUNCE, upload=False)   # AA on


def test_motion_blur_moves_the_primitives(ctx):
    """MOTION_BLUR: moveGeom every 4th iteration with dt = 0.10 (pathtrace.cu:318-331, 442-446); the reference's cornell.txt
    gives the sphere VEL 0 -0.1 0 (scenes/Scenes/cornell.txt:123)."""
    import oracle
    sc = _cornell((96, 72), 4)
    sc.geoms[-1].vel[:] = [0.0, -0.1, 0.0]
    W, H = 96, 72
    accum = np.zeros(3 * W * H, np.float32)
    g_ref = np.zeros((10, H, W), np.float32)
    refs = []
    moving = oracle.OracleScene.parse(CORNELL, res=(W, H), depth=4)
    moving.geoms[-1].vel[:] = [0.0, -0.1, 0.0]
    for it in range(1, 9):
        if it % 4 == 0:
            moving.move_geoms(0.10)
        moving.pathtrace(iter=it, accum=accum, gbuf=g_ref)
        refs.append(g_ref.copy())
    got = []
    gpu_trace(ctx, sc, 4, flags=api.TRACE_DEFAULT | api.TRACE_MOTION_BLUR, iters=8, each_iter=lambda it, g: got.append(g.copy()))
    for it in range(8):
        _same(got[it], refs[it], f"motion blur, iteration {it + 1}")
    still = []
    gpu_trace(ctx, sc, 4, iters=8, each_iter=lambda it, g: still.append(g.copy()))
    _same(still[2], got[2], "before the first move")
    assert not np.array_equal(still[7][0:3], got[7][0:3])


# ------------------------------------------------------------------ full benchmark sizes against the oracle (CPU BVH, same result)
def _full_size(ctx, sc, depth, what):
    import oracle
    g_ref, n_ref, m_ref = sc.pathtrace(flags=oracle.TRACE_AA | oracle.TRACE_COMPACT | oracle.TRACE_ORACLE_BVH)
    g, n, mt = gpu_trace(ctx, sc, depth)
    assert np.array_equal(mt, m_ref) and n[:len(n_ref)].tolist() == n_ref.tolist(), what
    _same(g, g_ref, what)
    return m_ref


def test_configs2_diffuse_sponza_1280x720_bit_exact(ctx):
    """BASELINE configs[2] as named: 262 144-triangle diffuse atrium, 1280x720, depth 8 -- every G-buffer bit against the oracle
    (whose CPU BVH is checked against the reference's exhaustive loop in tests/test_oracle_trace.py)."""
    sc = _cornell((1280, 720), 8)
    first = add_materials(sc, [synth.STONE])
    faces, lb, ub = synth.make_atrium_mesh(262144, 565, material=first)
    sc.set_mesh(faces, lb, ub)
    m = _full_size(ctx, sc, 8, "configs[2]")
    assert (m == first).sum() > 100000


def test_configs3_reflective_sponza_1280x720_bit_exact(ctx):
    sc = _cornell((1280, 720), 8)
    first = add_materials(sc, [synth.STONE, synth.MIRROR])
    faces, lb, ub = synth.make_atrium_mesh(262144, 565, material=first, floor_material=first + 1, column_material=first + 1)
    sc.set_mesh(faces, lb, ub)
    m = _full_size(ctx, sc, 8, "configs[3]")
    assert (m == first + 1).sum() > 100000


def test_configs4_living_room_1920x1080_depth12_bit_exact(ctx):
    sc = _living_room((1920, 1080), 12, 524288)
    m = _full_size(ctx, sc, 12, "configs[4]")
    assert {7, 8} <= set(np.unique(m).tolist())


# ---- the compile-time branches the reference ships switched off, as run-time flags (VERDICT r2 "what's missing" 1, 3, 4)

def _glassy_scene(res, depth, nfaces=4096):
    """Cornell + a small atrium whose floor is a mirror and whose columns are glass-with-reflection (REFL and REFR both set:
    the Glass_BxDF branch), a mirror sphere and a diffuse box: every branch of the DIELECTRIC scatterRay is taken"""
    sc = _cornell(res, depth)
    both = synth.material((.9, .95, 1.0), spec=(.98, .98, .98), refl=0.5, refr=1.0, ior=1.5)
    first = add_materials(sc, [synth.STONE, synth.MIRROR, synth.GLASS, both])
    faces, lb, ub = synth.make_atrium_mesh(nfaces, 565, material=first, floor_material=first + 1, column_material=first + 3)
    sc.set_mesh(faces, lb, ub)
    sc.geoms[-1].materialid = first + 2               # the sphere: refraction only
    return sc, first


def test_dielectric_branch_bit_exact(ctx):
    """AIPT_TRACE_DIELECTRIC = DIELECTRIC true (interactions.h:6, 88-168, 179-192)"""
    import oracle
    sc, first = _glassy_scene((160, 120), 8)
    fl = oracle.TRACE_AA | oracle.TRACE_COMPACT
    g_ref, n_ref, m_ref = sc.pathtrace(flags=fl | api.TRACE_DIELECTRIC)
    g, n, mt = gpu_trace(ctx, sc, 8, flags=api.TRACE_DEFAULT | api.TRACE_RECORD_MAT0 | api.TRACE_DIELECTRIC)
    assert np.array_equal(mt, m_ref) and n[:len(n_ref)].tolist() == n_ref.tolist()
    _same(g, g_ref, "dielectric branch")
    for k in (1, 2, 3):
        assert (m_ref == first + k).sum() > 100        # mirror, glass and glass-with-reflection surfaces are all visible
    g_plain, _, _ = sc.pathtrace(flags=fl)
    assert not np.array_equal(bits(g_plain), bits(g_ref))   # and the branch is a different renderer


def test_mesh_normal_view_bit_exact(ctx):
    """AIPT_TRACE_MESH_NORMAL_VIEW = MESH_NORMAL_VIEW true (interactions.h:4, 222-255)"""
    import oracle
    sc, first = _glassy_scene((128, 96), 6)
    fl = oracle.TRACE_AA | oracle.TRACE_COMPACT
    g_ref, n_ref, m_ref = sc.pathtrace(flags=fl | api.TRACE_MESH_NORMAL_VIEW)
    g, n, mt = gpu_trace(ctx, sc, 6, flags=api.TRACE_DEFAULT | api.TRACE_RECORD_MAT0 | api.TRACE_MESH_NORMAL_VIEW)
    assert n[:len(n_ref)].tolist() == n_ref.tolist()
    _same(g, g_ref, "normal view")


def test_no_cull_bit_exact(ctx):
    """AIPT_TRACE_NO_CULL = RAY_CULLING false (pathtrace.cu:23, 270-281): no scene-AABB test in front of the mesh; the BVH walk
    and the brute-force loop agree with the oracle's loop over all faces"""
    import oracle
    sc, first = _glassy_scene((128, 96), 6, nfaces=2048)
    fl = oracle.TRACE_AA | oracle.TRACE_COMPACT
    g_ref, n_ref, m_ref = sc.pathtrace(flags=fl | api.TRACE_NO_CULL)
    for extra in (0, api.TRACE_BRUTE_FORCE):
        g, n, mt = gpu_trace(ctx, sc, 6, flags=api.TRACE_DEFAULT | api.TRACE_RECORD_MAT0 | api.TRACE_NO_CULL | extra)
        assert np.array_equal(mt, m_ref) and n[:len(n_ref)].tolist() == n_ref.tolist()
        _same(g, g_ref, f"no cull ({extra})")
    # batches of frames take the flag too
    cams = [api.Camera.from_buffer_copy(bytes(sc.camera)) for _ in range(3)]
    import torch
    W, H = sc.camera.res[0], sc.camera.res[1]
    ctx.trace_configure_batch(W, H, 3)
    gb = torch.zeros(3, 10, H, W, device="cuda")
    torch.cuda.synchronize()
    ctx.pathtrace_batch(cams, 1, 6, gb, api.TRACE_DEFAULT | api.TRACE_NO_CULL)
    ctx.sync()
    for k in range(3):
        _same(gb[k].cpu().numpy(), g_ref, f"no cull, batch frame {k}")
