// trace.hip -- 1-spp wavefront path tracer for CDNA4 (gfx950), hand-written HIP.
//
// Replaces pathtraceInit / pathtrace / pathtraceFree (reference Inference/src/pathtrace.h:6-8, pathtrace.cu:96-145,
// 422-528) and the device math of intersections.h / interactions.h.  Results are defined by the reference's kernels
// (generateRayFromCamera :155-182, computeIntersections :200-306, shadeMaterial :333-390, thrust::partition :505,
// finalGather :393-402, copy_data :81-94); the execution plan is not:
//
//  * Path state is three float4 planes in HBM (48 B/path: origin+dx | dy dz + colour rg | colour b + remaining bounces) and
//    NEVER MOVES: path i is pixel i for the whole frame.  A bounce loads and stores it with three 16-byte accesses per lane
//    (what bounds this kernel is the number of per-lane vector-memory accesses, see bvh.cpp and DESIGN.md).
//  * Stream compaction is logical.  thrust::partition keeps survivors in their original relative order, so the index
//    a live path has after compaction -- which seeds its RNG, pathtrace.cu:351 -- is its RANK among live paths in array
//    order.  trace_compact turns the per-slot alive flags of a bounce into the next bounce's live list (workgroup prefix +
//    wave64 __ballot/__popcll prefix).  No scan kernel, no host sync per bounce (the reference syncs twice per bounce).
//  * One kernel per bounce fuses ray generation (bounce 0), nearest-hit search, shading/scatter, the G-buffer
//    writes (planes 3-9, already h-flipped) and finalGather+copy_data: a path deposits its colour into planes 0-2
//    the moment its remainingBounces reaches 0 (light, miss, or depth exhausted -- SURVEY F9).
//  * Mesh: a 4-wide BVH with 8-bit child boxes (64-byte nodes) walked front to back on a per-lane LDS stack, 48-byte leaf
//    triangle records, the winning face fetched once after the walk (bvh.cpp).  A leaf's triangles are tested by all 64 lanes
//    of the wave (coop_leaf_step); idle lanes take over subtrees of the busy lanes' walks, all parts of a ray meeting in one
//    (t, face index) minimum (steal_step): the result never depends on which lane walked what.  (Rounds 2-5 also pooled a
//    workgroup's rays in batched traces and refilled idle lanes from the pool; with the split walks that was worth +0.5 % and
//    left in round 6: 600 lines, 19 KB of LDS -- tools/experiments/README.md.)
//
// Arithmetic is plain IEEE fp32 (this file is compiled with -ffp-contract=off, correctly rounded divide/sqrt), in the
// statement order of the reference/GLM sources, so the output matches the CPU restatement used by the tests bit for
// bit.  sin/cos of the hemisphere angle use the same Cody-Waite + minimax polynomial on both sides (det_sincosf).
#include "internal.h"

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <cstring>

namespace aipt {

struct v3 { float x, y, z; };

struct DevGeom {          // one primitive; matrices column-major (glm::mat4)
    int type, materialid;
    float inv[16], xf[16], invT[16];
    float lo[3], hi[3];   // padded world-space box around the primitive (broad phase only, never decides a hit)
};
constexpr int MAXG_LDS = 32;   // scenes with up to this many primitives use the per-lane candidate loop
constexpr int MAXM_LDS = 32;   // materials kept in LDS (more: read from HBM)
struct DevFace {          // Face, sceneStructs.h:40
    float v[3][3], n[3][3];
    int materialid;
};
struct alignas(16) DevFaceP { DevFace f; int pad; };   // 80 bytes: five 16-byte loads (leaf order, fetched once per ray)
static_assert(sizeof(DevFaceP) == 80, "DevFaceP");

constexpr int MAX_DEPTH = 64;
constexpr int BMAX = 24;       // most frames one batched trace (aipt_trace_batch) can hold

struct TraceState {
    DevGeom* d_geoms = nullptr; int ngeoms = 0;
    aipt_material* d_mats = nullptr; int nmats = 0;
    DevFace* d_faces = nullptr; int nfaces = 0;      // caller's order (AIPT_TRACE_BRUTE_FORCE)
    int stack_need = 0;
    Bvh4Node* d_nodes = nullptr; int nnodes = 0;     // 4-wide BVH over the faces (bvh.cpp)
    TriRec* d_tris = nullptr;                        // leaf-ordered triangle records
    DevFaceP* d_lfaces = nullptr;                    // leaf-ordered full faces
    int* d_fslot = nullptr;                          // face index (caller's order) -> leaf slot
    aipt_aabb box{};
    bool have_scene = false;
    int W = 0, H = 0, P = 0, nblk = 0;
    int batch = 1;                // frames the per-path buffers below can hold (aipt_trace_configure_batch): sized batch * P
    aipt_camera* d_cams = nullptr;   // [BMAX] cameras of a batched trace
    int* d_nlive_f = nullptr;     // [MAX_DEPTH+1][BMAX] live paths per bounce and frame
    int* d_cntf = nullptr;        // [workgroups][BMAX] per-frame survivor counts of the last bounce
    int* d_rank[2] = {nullptr, nullptr};   // per-frame ranks of the live list entries (batched traces), ping-pong like d_live
    int last_frames = 1;
    float4* d_state = nullptr;    // [3][P]: (ox oy oz dx) (dy dz cr cg) (cb rem . .)
    int* d_cnt = nullptr;         // per-workgroup live counts of the last bounce
    int* d_alive = nullptr;       // [P] per array slot: does the path survive the bounce?
    int* d_live[3] = {nullptr, nullptr, nullptr};  // live-path index lists (third one: AIPT_TRACE_SORT_MATERIAL)
    int* d_nlive = nullptr;       // [MAX_DEPTH+1]
    int* d_mat0 = nullptr;        // [P]
    float* d_image = nullptr;     // [3][P] radiance accumulated over iterations 1..n (dev_image, pathtrace.cu:101), h-flipped
    float* d_cache = nullptr;     // [8][P] bounce-0 hit records of iteration 1 (AIPT_TRACE_CACHE_FIRST_BOUNCE): t, material, P, raw N
    int* d_sortkey = nullptr;     // [P] material id of the hit found in array slot t at the current bounce (0 = miss)
    int* d_hist = nullptr;        // [nkeys][nblk] counting-sort histogram / bases
    int* d_stack_ovf = nullptr; int ovf_entries = 0;   // traversal-stack overflow, [entries][P]
    int hist_keys = 0;
    bool cache_valid = false;
    std::vector<aipt_geom> h_geoms;   // host copy for AIPT_TRACE_MOTION_BLUR (moveGeom, pathtrace.cu:318-331)
    int last_depth = 0;
    bool mat0_valid = false;
    // HIP-event timing of the bounce launches (aipt_trace_profile_*)
    int prof_max = 0, prof_calls = 0, prof_every = 1, prof_seen = 0;
    std::vector<hipEvent_t> prof_ev;              // [call][bounce][2]
    std::vector<int> prof_frames;                 // [call] frames the recorded call held
    char kname[2][40] = {};                       // instantiation that ran bounce 0 / the later bounces in the last trace
    // Second LANE of per-path buffers (lane 1 of trace_on_stream): aipt_frames traces the two halves of a call's frames BESIDE
    // each other on two streams (every bounce launch ends in a tail -- its longest rays' chains of dependent node visits -- and
    // two launch sequences side by side fill each other's tails).  The lane shares the scene
    // (pointers copied at every trace) and owns only its path state, sized for half of the batch.
    TraceState* side = nullptr;
    bool is_side = false;
    bool side_used = false;                       // the last trace of this lane was followed by one on the side lane (a two-lane call)
};

struct TraceParams {
    aipt_camera cams[BMAX];      // cameras of the frames traced together (kernel argument; cams[0] for a single frame)
    int nframes;                 // frames traced together: path i = frame i / P, pixel i % P
    int PT;                      // nframes * P paths
    size_t PS;                   // plane stride of the per-path buffers (their capacity, batch * P)
    size_t gbuf_frame;           // floats between the G-buffers of consecutive frames
    int* n_live_f;               // [MAX_DEPTH+1][BMAX] live paths per bounce and frame (batched: the RNG index is the rank
                                 // among the live paths of the SAME frame, as an unbatched trace of that frame computes it)
    int iter, trace_depth, bounce;
    uint32_t flags;
    int W, H, P;
    float4* st;
    const DevGeom* geoms; int ngeoms;
    const aipt_material* mats; int nmats;
    const DevFace* faces; int nfaces;
    const Bvh4Node* nodes;
    const TriRec* tris;
    const DevFaceP* lfaces;
    const int* fslot;            // leaf slot of a face index (the split walk's results are (distance, face index) keys)
    aipt_aabb box;
    float* gbuf; size_t plane; int stride;
    int* cnt;                    // per-workgroup live counts (bounce -> compact)
    int* alive;                  // per array slot: survives this bounce (bounce -> compact)
    int* n_live;
    int* mat0;
    float* image;
    const int* live_in;          // pixel indices of the live paths entering this bounce, in array order (nullptr: all pixels)
    int* live_out;
    float* cache; int cache_mode;   // 0: none, 1: save the bounce-0 hit records, 2: reuse them instead of intersecting
    int* sortkey;                   // AIPT_TRACE_SORT_MATERIAL: material id of the hit in array slot t (0 = miss)
    int* hist; int nkeys, nblk;
    const int* sort_in; int* sort_out;
    int* stack_ovf;                 // [stack bound - STACK_LDS][P] traversal-stack overflow (see WalkStack)
    // batched trace: per-frame survivor counts of every workgroup (bounce -> compact) and the per-frame rank of every live
    // list entry (compact -> next bounce)
    int* cntf;                      // [workgroups][BMAX]
    const int* rank_in; int* rank_out;
};

// ---------------------------------------------------------------------------------------------- vector helpers
__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 vcross(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
__device__ __forceinline__ float vlength(v3 a) { return sqrtf(vdot(a, a)); }
__device__ __forceinline__ v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }
__device__ __forceinline__ v3 vreflect(v3 I, v3 N) { return vsub(I, vscale(vscale(N, vdot(N, I)), 2.0f)); }
__device__ __forceinline__ v3 glm_refract(v3 I, v3 N, float eta) {          // func_geometric.inl:191-200
    const float d = vdot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    const v3 r = vsub(vscale(I, eta), vscale(N, eta * d + sqrtf(k)));
    return vscale(r, (float)(k >= 0.0f));
}
__device__ __forceinline__ float glm_min(float x, float y) { return x < y ? x : y; }
__device__ __forceinline__ float glm_max(float x, float y) { return x > y ? x : y; }
__device__ __forceinline__ v3 mulMV(const float* m, v3 v, float w) {        // type_mat4x4.inl:618-629
    v3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * w);
    r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * w);
    r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * w);
    return r;
}

// ---------------------------------------------------------------------------------------------- RNG
__device__ __forceinline__ uint32_t utilhash(uint32_t a) {                  // intersections.h:12-20
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
// x mod (2^31-1) for x < 2^63 via the Mersenne fold (exact)
__device__ __forceinline__ uint32_t mod_m31(uint64_t p) {
    uint32_t r = (uint32_t)(p & 0x7fffffffu) + (uint32_t)(p >> 31);
    return r >= 2147483647u ? r - 2147483647u : r;
}
__device__ __forceinline__ uint32_t make_seed(int iter, int index, int depth) {   // pathtrace.cu:52-56 + LCG seed()
    const uint32_t h = utilhash((1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter) ^ utilhash((uint32_t)index);
    const uint32_t s = mod_m31(h);
    return s == 0 ? 1u : s;
}
__device__ __forceinline__ float u01(uint32_t& x, float a, float b) {       // minstd_rand + uniform_real_distribution<float>
    x = mod_m31((uint64_t)x * 48271ull);
    float r = (float)(x - 1u);
    r /= (1.0f + (float)(2147483646u - 1u));
    return (r * (b - a)) + a;
}
__device__ __forceinline__ void det_sincosf(float x, float& s, float& c) {
    const int q = (int)(x * 0.636619772367581343f + 0.5f);
    const float fq = (float)q;
    float r = x - fq * 1.5703125f;
    r = r - fq * 4.837512969970703125e-4f;
    r = r - fq * 7.54978995489188216e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
                     - 0.5f * z + 1.0f;
    switch (q & 3) {
        case 0: s = ps; c = pc; break;
        case 1: s = pc; c = -ps; break;
        case 2: s = -ps; c = -pc; break;
        default: s = -pc; c = ps; break;
    }
}

// ---------------------------------------------------------------------------------------------- intersections.h
__device__ __forceinline__ v3 getPointOnRay(v3 o, v3 d, float t) { return vadd(o, vscale(vnormalize(d), t - .0001f)); }

__device__ __forceinline__ float boxTest(const DevGeom& g, v3 ro, v3 rd, v3& P, v3& N) {     // :52-94
    const v3 qo = mulMV(g.inv, ro, 1.0f);
    const v3 qd = vnormalize(mulMV(g.inv, rd, 0.0f));
    float tmin = -1e38f, tmax = 1e38f;
    v3 tmin_n = V(0, 0, 0), tmax_n = V(0, 0, 0);
    const float o[3] = {qo.x, qo.y, qo.z}, d[3] = {qd.x, qd.y, qd.z};
#pragma unroll
    for (int xyz = 0; xyz < 3; ++xyz) {
        const float t1 = (-0.5f - o[xyz]) / d[xyz];
        const float t2 = (+0.5f - o[xyz]) / d[xyz];
        const float ta = glm_min(t1, t2), tb = glm_max(t1, t2);
        const float sgn = t2 < t1 ? +1.0f : -1.0f;
        const v3 n = V(xyz == 0 ? sgn : 0.0f, xyz == 1 ? sgn : 0.0f, xyz == 2 ? sgn : 0.0f);
        if (ta > 0 && ta > tmin) { tmin = ta; tmin_n = n; }
        if (tb < tmax) { tmax = tb; tmax_n = n; }
    }
    if (tmax >= tmin && tmax > 0) {
        if (tmin <= 0) { tmin = tmax; tmin_n = tmax_n; }
        P = mulMV(g.xf, getPointOnRay(qo, qd, tmin), 1.0f);
        N = vnormalize(mulMV(g.xf, tmin_n, 0.0f));
        return vlength(vsub(ro, P));
    }
    return -1.0f;
}

__device__ __forceinline__ float sphereTest(const DevGeom& g, v3 ro, v3 rd, v3& P, v3& N) {  // :106-148
    const v3 o = mulMV(g.inv, ro, 1.0f);
    const v3 d = vnormalize(mulMV(g.inv, rd, 0.0f));
    const float vDotDirection = vdot(o, d);
    const float radicand = vDotDirection * vDotDirection - (vdot(o, o) - 0.5f * 0.5f);
    if (radicand < 0) return -1.0f;
    const float squareRoot = sqrtf(radicand);
    const float firstTerm = -vDotDirection;
    const float t1 = firstTerm + squareRoot, t2 = firstTerm - squareRoot;
    float t;
    bool outside;
    if (t1 < 0 && t2 < 0) return -1.0f;
    else if (t1 > 0 && t2 > 0) { t = fminf(t1, t2); outside = true; }
    else { t = fmaxf(t1, t2); outside = false; }
    const v3 obj = getPointOnRay(o, d, t);
    P = mulMV(g.xf, obj, 1.0f);
    N = vnormalize(mulMV(g.invT, obj, 0.0f));
    if (!outside) N = vneg(N);
    return vlength(vsub(ro, P));
}

__device__ float triangleTest(const DevFace& f, v3 orig, v3 dir, v3& P, v3& N) {   // :159-172 + gtx/intersect.inl:37-74
    const v3 v0 = V(f.v[0][0], f.v[0][1], f.v[0][2]);
    const v3 v1 = V(f.v[1][0], f.v[1][1], f.v[1][2]);
    const v3 v2 = V(f.v[2][0], f.v[2][1], f.v[2][2]);
    const v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    const v3 p = vcross(dir, e2);
    const float a = vdot(e1, p);
    if (a < FLT_EPSILON) return -1.0f;
    const float ff = 1.0f / a;
    const v3 s = vsub(orig, v0);
    const float bx = ff * vdot(s, p);
    if (bx < 0.0f) return -1.0f;
    if (bx > 1.0f) return -1.0f;
    const v3 q = vcross(s, e1);
    const float by = ff * vdot(dir, q);
    if (by < 0.0f) return -1.0f;
    if (by + bx > 1.0f) return -1.0f;
    const float bz = ff * vdot(e2, q);
    if (!(bz >= 0.0f)) return -1.0f;
    const float bw = 1.0f - bx - by;
    P = vadd(vadd(vscale(v0, bx), vscale(v1, by)), vscale(v2, bw));       // SURVEY F8, reproduced on purpose
    const v3 n0 = V(f.n[0][0], f.n[0][1], f.n[0][2]);
    const v3 n1 = V(f.n[1][0], f.n[1][1], f.n[1][2]);
    const v3 n2 = V(f.n[2][0], f.n[2][1], f.n[2][2]);
    N = vnormalize(vadd(vadd(vscale(n0, bw), vscale(n1, bx)), vscale(n2, by)));
    return bz;
}

// Broad phase: can the ray touch the primitive's padded world box at all?  Conservative by construction -- the box is
// padded by 1e-3 of its scale on the host, the slab arithmetic here is good to 3e-7 relative, NaNs answer "maybe" -- so
// a primitive it rejects is one whose exact test returns "no hit", and skipping that test changes no result bit.
// tn receives the distance at which the ray enters the padded box (a lower bound of any hit distance the exact test can
// return for this primitive: the exact hit point lies inside the padded box).
__device__ __forceinline__ bool maybe_hits(const float* lo, const float* hi, v3 o, v3 inv, float& tn) {
    const float t1 = (lo[0] - o.x) * inv.x, t2 = (hi[0] - o.x) * inv.x;
    const float t3 = (lo[1] - o.y) * inv.y, t4 = (hi[1] - o.y) * inv.y;
    const float t5 = (lo[2] - o.z) * inv.z, t6 = (hi[2] - o.z) * inv.z;
    tn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    const float tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    return !(tf < 0.0f || tn > tf);
}

__device__ bool rayAABB(v3 ro, v3 rd, const aipt_aabb& bb) {                  // :175-200
    const float dx = 1.0f / rd.x, dy = 1.0f / rd.y, dz = 1.0f / rd.z;
    const float t1 = (bb.lb[0] - ro.x) * dx, t2 = (bb.ub[0] - ro.x) * dx;
    const float t3 = (bb.lb[1] - ro.y) * dy, t4 = (bb.ub[1] - ro.y) * dy;
    const float t5 = (bb.lb[2] - ro.z) * dz, t6 = (bb.ub[2] - ro.z) * dz;
    const float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    const float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    if (tmax < 0) return false;
    if (tmin > tmax) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------- interactions.h
__device__ v3 hemisphere(v3 normal, uint32_t& rng) {                          // :13-44
    const float up = sqrtf(u01(rng, 0.0f, 1.0f));
    const float over = sqrtf(1 - up * up);
    const float around = u01(rng, 0.0f, 1.0f) * 6.2831853071795864769252867665590057683943f;
    v3 dnn;
    if (fabsf(normal.x) < 0.5773502691896257645091487805019574556476f) dnn = V(1, 0, 0);
    else if (fabsf(normal.y) < 0.5773502691896257645091487805019574556476f) dnn = V(0, 1, 0);
    else dnn = V(0, 0, 1);
    const v3 p1 = vnormalize(vcross(normal, dnn));
    const v3 p2 = vnormalize(vcross(normal, p1));
    float sn, cs;
    det_sincosf(around, sn, cs);
    return vadd(vadd(vscale(normal, up), vscale(p1, cs * over)), vscale(p2, sn * over));
}
__device__ __forceinline__ bool ref_refract(v3 v, v3 n, float ni_over_nt, v3& refracted) {   // :74-85
    const v3 uv = vnormalize(v);
    const float dt = vdot(uv, n);
    const float discriminat = (float)(1.0 - (double)(ni_over_nt * ni_over_nt * (1 - dt * dt)));
    if (discriminat > 0) {
        refracted = vsub(vscale(vsub(uv, vscale(n, dt)), ni_over_nt), vscale(n, sqrtf(discriminat)));
        return true;
    }
    return false;
}
__device__ __forceinline__ float schlick(float cosine, float ref_idx) {       // :116-120
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    const float x = 1 - cosine;
    const float x2 = x * x;
    return r0 + (1 - r0) * ((x2 * x2) * x);
}

// ---- the DIELECTRIC branch of interactions.h (:6 false in the reference; AIPT_TRACE_DIELECTRIC here): :88-168, :179-192
__device__ __forceinline__ float fresnelDielectric(float cosThetaI, float etaI, float etaT) {   // :88-115
    cosThetaI = glm_min(glm_max(cosThetaI, -1.0f), 1.0f);                     // glm::clamp
    const bool entering = cosThetaI > 0.0f;
    float etaIb = etaI, etaTb = etaT;
    if (!entering) { etaIb = etaT; etaTb = etaI; cosThetaI = fabsf(cosThetaI); }
    const float sinThetaI = sqrtf(glm_max(0.0f, 1 - cosThetaI * cosThetaI));
    const float sinThetaT = etaIb / etaTb * sinThetaI;
    if (sinThetaT >= 1) return 1.0f;
    const float cosThetaT = sqrtf(glm_max(0.0f, 1 - sinThetaT * sinThetaT));
    const float Rparl = ((etaTb * cosThetaI) - (etaIb * cosThetaT)) / ((etaTb * cosThetaI) + (etaIb * cosThetaT));
    const float Rperp = ((etaIb * cosThetaI) - (etaTb * cosThetaT)) / ((etaIb * cosThetaI) + (etaTb * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
__device__ __forceinline__ void specularReflection(v3& origin, v3& direction, v3& pcolor, v3 hitP, v3 normal, const aipt_material& m) {   // :121-125
    pcolor = vmul(pcolor, V(m.specular_color[0], m.specular_color[1], m.specular_color[2]));
    direction = vreflect(direction, normal);
    origin = vadd(hitP, vscale(direction, .001f));
}
__device__ __forceinline__ void specularRefraction(v3& origin, v3& direction, v3& pcolor, v3 hitP, v3 normal, const aipt_material& m) {   // :127-146
    const v3 wo = direction;
    const bool leaving = vdot(wo, normal) > 0.f;
    const v3 n = vscale(normal, leaving ? -1.f : 1.f);
    const float eta = leaving ? m.indexOfRefraction : (1.f / m.indexOfRefraction);
    v3 wi = glm_refract(wo, n, eta);
    if (vlength(wi) < .01f) {                                                 // total internal reflection
        pcolor = vscale(pcolor, 0.0f);
        wi = vreflect(wo, normal);
    }
    pcolor = vmul(pcolor, V(m.specular_color[0], m.specular_color[1], m.specular_color[2]));
    direction = wi;
    origin = vadd(hitP, vscale(direction, .001f));
}
__device__ void scatterDielectric(v3& origin, v3& direction, v3& pcolor, v3 hitN, v3 hitP, const aipt_material& m, uint32_t& rng) {   // :179-192
    const float REF_EPSILON = 0.0001f;                                        // utilities.h:16
    if (m.hasReflective > REF_EPSILON && m.hasRefractive > REF_EPSILON) {     // Glass_BxDF :148-163
        const float VdotN = vdot(vneg(direction), hitN);
        const bool leaving = VdotN < 0.f;
        const float eI = leaving ? m.indexOfRefraction : 1.f;
        const float eT = leaving ? 1.f : m.indexOfRefraction;
        const float fresnel = fresnelDielectric(VdotN, eI, eT) / fabsf(VdotN);
        if (u01(rng, 0.0f, 1.0f) < fresnel) specularReflection(origin, direction, pcolor, hitP, hitN, m);
        else specularRefraction(origin, direction, pcolor, hitP, hitN, m);
    } else if (m.hasReflective > REF_EPSILON) {
        specularReflection(origin, direction, pcolor, hitP, hitN, m);
    } else if (m.hasRefractive > REF_EPSILON) {
        specularRefraction(origin, direction, pcolor, hitP, hitN, m);
    } else {                                                                  // Lambert_BxDF :164-168
        direction = hemisphere(vnormalize(hitN), rng);
        pcolor = vmul(pcolor, V(m.color[0], m.color[1], m.color[2]));
        origin = vadd(hitP, vscale(direction, .001f));
    }
}

// scatterRay: the live branch (DIELECTRIC false, FRESNELS true, :194-258), the DIELECTRIC branch under AIPT_TRACE_DIELECTRIC,
// MESH_NORMAL_VIEW (:4, :222-255: the surface normal as the colour) under AIPT_TRACE_MESH_NORMAL_VIEW
__device__ void scatterRay(v3& origin, v3& direction, v3& pcolor, v3 hitN, v3 hitP, const aipt_material& m, uint32_t& rng, uint32_t flags) {
    if (flags & AIPT_TRACE_DIELECTRIC) { scatterDielectric(origin, direction, pcolor, hitN, hitP, m, rng); return; }
    const bool nview = (flags & AIPT_TRACE_MESH_NORMAL_VIEW) != 0;
    v3 dir = direction;
    v3 color;
    const v3 mcolor = nview ? hitN : V(m.color[0], m.color[1], m.color[2]);
    const v3 scolor = nview ? hitN : V(m.specular_color[0], m.specular_color[1], m.specular_color[2]);
    float reflective_prob = m.hasReflective;
    if (reflective_prob != 0 || m.hasRefractive != 0) {
        const float pdf = u01(rng, 0.0f, 1.0f);
        float refrac_index_ratio, cosine;
        v3 normal;
        cosine = vdot(vnormalize(dir), hitN);
        if (cosine <= 0) {
            normal = hitN;
            refrac_index_ratio = 1 / m.indexOfRefraction;
            cosine = -cosine;
        } else {
            normal = vneg(hitN);
            refrac_index_ratio = m.indexOfRefraction;
        }
        if (ref_refract(direction, normal, refrac_index_ratio, dir))   // overwrites dir, as the reference does
            reflective_prob = schlick(cosine, refrac_index_ratio);
        else
            reflective_prob = 1.0f;
        if (pdf < reflective_prob) {
            dir = vnormalize(vreflect(dir, hitN));
            color = scolor;
        } else {
            dir = vnormalize(glm_refract(direction, normal, refrac_index_ratio));
            if (!vlength(dir)) {
                dir = vnormalize(vreflect(dir, hitN));
                color = scolor;
            } else
                color = mcolor;
        }
    } else {
        dir = vnormalize(hemisphere(hitN, rng));
        color = mcolor;
    }
    direction = dir;
    origin = vadd(hitP, vscale(dir, 0.01f));
    if (nview) color = V(fabsf(color.x), fabsf(color.y), fabsf(color.z));     // :254
    pcolor = vmul(pcolor, color);
}

// glm::intersectRayTriangle (gtx/intersect.inl:37-74) on a leaf record: e1, e2 are the subtractions the function starts with,
// done once on the host in fp32.  Returns the hit distance (baryPosition.z) or -1; same operations in the same order as
// triangleTest above, so the value is the one the full test of that face returns.
__device__ __forceinline__ float triHitT(v3 v0, v3 e1, v3 e2, v3 orig, v3 dir) {
    const v3 p = vcross(dir, e2);
    const float a = vdot(e1, p);
    if (a < FLT_EPSILON) return -1.0f;
    const float ff = 1.0f / a;
    const v3 s = vsub(orig, v0);
    const float bx = ff * vdot(s, p);
    if (bx < 0.0f) return -1.0f;
    if (bx > 1.0f) return -1.0f;
    const v3 q = vcross(s, e1);
    const float by = ff * vdot(dir, q);
    if (by < 0.0f) return -1.0f;
    if (by + bx > 1.0f) return -1.0f;
    const float bz = ff * vdot(e2, q);
    if (!(bz >= 0.0f)) return -1.0f;
    return bz;
}

// -DAIPT_TRACE_STATS: walk statistics for tools/trace_stats.py (lane-level node visits and triangle tests, wave-level loop
// trips); compiled out of the product build.
#if defined(AIPT_TRACE_PHASES) && !defined(AIPT_TRACE_STATS)      // phase stamps alone: the walk counters perturb the timing
#define AIPT_TRACE_STATS
#define AIPT_TRACE_NO_COUNTERS
#endif
#ifdef AIPT_TRACE_STATS
__device__ unsigned long long g_trace_stats[16];
#ifdef AIPT_TRACE_NO_COUNTERS
#define STAT_ADD(k, v) do {} while (0)
#else
#define STAT_ADD(k, v) atomicAdd(&g_trace_stats[k], (unsigned long long)(v))
#endif
__device__ unsigned long long g_phase[8];
#define PHASE(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_phase[k], now_ - ph_t); ph_t = now_; } while (0)
#define PHASE_INIT() unsigned long long ph_t = __builtin_amdgcn_s_memtime()
#ifdef AIPT_TRACE_NO_COUNTERS
#define STAT_WAVE(k) do {} while (0)
#else
#define STAT_WAVE(k) do { if (__ffsll((long long)__ballot(1)) - 1 == (int)(threadIdx.x & 63)) atomicAdd(&g_trace_stats[k], 1ull); } while (0)
#endif
#else
#define STAT_ADD(k, v) do {} while (0)
#define STAT_WAVE(k) do {} while (0)
#define PHASE(k) do {} while (0)
#define PHASE_INIT() do {} while (0)
#endif

__device__ __forceinline__ float ubyte_f(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }   // v_cvt_f32_ubyteK
__device__ __forceinline__ void cas(float& ka, int& ra, float& kb, int& rb) {      // compare-exchange: (ka, ra) <= (kb, rb)
    const bool sw = ka > kb;
    const float tk = sw ? kb : ka, uk = sw ? ka : kb;
    const int tr = sw ? rb : ra, ur = sw ? ra : rb;
    ka = tk; kb = uk; ra = tr; rb = ur;
}

// glm::intersectRayTriangle on a leaf record, straight-line: every comparison of the reference is evaluated and combined
// at the end (same operations, same operands, same value as triHitT / the reference's early returns; a lone wave pays more
// for a divergent branch than for the few operations it would skip).
__device__ __forceinline__ float triHitT_flat(v3 v0, v3 e1, v3 e2, v3 orig, v3 dir) {
    const v3 p = vcross(dir, e2);
    const float a = vdot(e1, p);
    const float ff = 1.0f / a;
    const v3 s = vsub(orig, v0);
    const float bx = ff * vdot(s, p);
    const v3 q = vcross(s, e1);
    const float by = ff * vdot(dir, q);
    const float bz = ff * vdot(e2, q);
    const bool miss = (a < FLT_EPSILON) | (bx < 0.0f) | (bx > 1.0f) | (by < 0.0f) | (by + bx > 1.0f) | !(bz >= 0.0f);
    return miss ? -1.0f : bz;
}

// Per-lane traversal stack: the first STACK_LDS entries live in LDS ([entry][thread]: conflict-free), deeper entries -- the
// builder's bound can reach BVH_MAX_STACK, the atrium never exceeds 14 and goes past 8 on 0.03 % of its node visits -- in a
// global overflow area ([entry][array slot]).  A full-size LDS stack would cap the kernel at 3 waves per SIMD.
constexpr int STACK_LDS = 8;
struct WalkStack {
    int* lds; int* ovf; size_t ostride; int sp;
    int bot;                        // entries [bot, sp) are live: the split walk gives its bottom entries away (steal_step)
    __device__ __forceinline__ void push(int v) {
        if (sp < STACK_LDS) lds[sp * 256] = v; else ovf[(size_t)(sp - STACK_LDS) * ostride] = v;
        sp++;
    }
    __device__ __forceinline__ bool empty() const { return sp == bot; }
    __device__ __forceinline__ int pop() {
        sp--;
        const int v = sp < STACK_LDS ? lds[sp * 256] : ovf[(size_t)(sp - STACK_LDS) * ostride];
        if (sp == bot) { sp = 0; bot = 0; }
        return v;
    }
};

// Nearest mesh hit through the 4-wide BVH.  Same result as the reference's loop over all faces in index order
// (pathtrace.cu:258-269): the reference's triangle test on the candidate faces; among equal distances the face with the lowest
// index wins and a face never replaces a primitive hit at equal distance (strict t_min > t in the reference's loop).
// Walk: "while-while" -- descend through inner nodes until the lane holds a leaf (or nothing), then test the leaf's
// triangles (loaded two at a time: one memory round trip per pair); the nearer children are visited first, the others wait on
// the per-lane stack.  Box test per child: the ray's direction signs pick the entry and the exit plane of every axis once per
// node (on the packed 4-child words), then t = fma(q, scale/d, (p - o)/d) on the 8-bit coordinate q, max3 / min3 -- 14 VALU
// operations per child; the fma's error corresponds to ~1e-6 in space, far inside the boxes' padding.
struct WalkRay {
    v3 o, d;
    float ix, iy, iz;
    bool negx, negy, negz;
    float t_min;
    int best_face, best_slot;
    __device__ __forceinline__ void start(v3 o_, v3 d_, float t_bound) {
        o = o_; d = d_;
        ix = 1.0f / d.x; iy = 1.0f / d.y; iz = 1.0f / d.z;
        negx = __float_as_uint(ix) >> 31; negy = __float_as_uint(iy) >> 31; negz = __float_as_uint(iz) >> 31;
        t_min = t_bound; best_face = -1; best_slot = -1;
    }
};
constexpr int WALK_DONE = BVH_EMPTY;

// one inner-node visit: box tests of the four children, nearest-first order, the nearest becomes `cur`, the others wait on
// the stack (or `cur` pops the stack / becomes WALK_DONE when no child is hit)
__device__ __forceinline__ void walk_node(const uint4* nodes, const WalkRay& r, WalkStack& st, int& cur) {
    const uint4* nd = nodes + (unsigned)cur * 4u;
    const uint4 w0 = nd[0], w1 = nd[1], w2 = nd[2], w3 = nd[3];
    const float sx = __uint_as_float((w0.w & 0xffu) << 23) * r.ix, sy = __uint_as_float(((w0.w >> 8) & 0xffu) << 23) * r.iy,
                sz = __uint_as_float(((w0.w >> 16) & 0xffu) << 23) * r.iz;
    const float bx = (__uint_as_float(w0.x) - r.o.x) * r.ix, by = (__uint_as_float(w0.y) - r.o.y) * r.iy,
                bz = (__uint_as_float(w0.z) - r.o.z) * r.iz;
    // entry / exit planes of the four children, per axis (qlo = w1.xyz, qhi = w1.w, w2.x, w2.y)
    const uint32_t nx = r.negx ? w1.w : w1.x, fx = r.negx ? w1.x : w1.w;
    const uint32_t ny = r.negy ? w2.x : w1.y, fy = r.negy ? w1.y : w2.x;
    const uint32_t nz = r.negz ? w2.y : w1.z, fz = r.negz ? w1.z : w2.y;
    float key[4];
    int ref[4] = {(int)w2.z, (int)w2.w, (int)w3.x, (int)w3.y};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float n = fmaxf(fmaxf(__builtin_fmaf(ubyte_f(nx, k), sx, bx), __builtin_fmaf(ubyte_f(ny, k), sy, by)),
                              __builtin_fmaf(ubyte_f(nz, k), sz, bz));
        const float f = fminf(fminf(__builtin_fmaf(ubyte_f(fx, k), sx, bx), __builtin_fmaf(ubyte_f(fy, k), sy, by)),
                              __builtin_fmaf(ubyte_f(fz, k), sz, bz));
        const bool h = ref[k] != BVH_EMPTY && !(f < 0.0f || n > f || n > r.t_min);    // NaN -> visit
        key[k] = h ? fmaxf(n, -FLT_MAX) : INFINITY;                                    // a NaN key of a hit sorts first
    }
    cas(key[0], ref[0], key[1], ref[1]);
    cas(key[2], ref[2], key[3], ref[3]);
    cas(key[0], ref[0], key[2], ref[2]);
    cas(key[1], ref[1], key[3], ref[3]);
    cas(key[1], ref[1], key[2], ref[2]);
    if (__builtin_expect(st.sp + 3 > STACK_LDS, 0)) {   // rare: some of the pushes may go to the overflow area
        if (key[3] < INFINITY) st.push(ref[3]);
        if (key[2] < INFINITY) st.push(ref[2]);
        if (key[1] < INFINITY) st.push(ref[1]);
    } else {                                            // farthest first: the stack pops the nearest
        if (key[3] < INFINITY) st.lds[st.sp++ * 256] = ref[3];
        if (key[2] < INFINITY) st.lds[st.sp++ * 256] = ref[2];
        if (key[1] < INFINITY) st.lds[st.sp++ * 256] = ref[1];
    }
#if defined(AIPT_TRACE_STATS) && !defined(AIPT_TRACE_NO_COUNTERS)
    atomicMax(&g_trace_stats[6], (unsigned long long)st.sp);
    if (st.sp > 8) STAT_ADD(7, 1);
    if (st.sp > 12) STAT_ADD(15, 1);
#endif
    if (key[0] < INFINITY) cur = ref[0];
    else cur = st.empty() ? WALK_DONE : st.pop();
}

// Cooperative leaf step.  The lanes of a wave that hold a leaf (5.2 triangles each on the atrium) list their (ray, triangle) pairs in LDS and ALL 64 lanes test one pair each per round: the owner's ray comes over by
// lane permutes, the result goes to the owner's slot by a 64-bit LDS atomicMin on (t bits, face index) -- the reference's
// "first strictly smaller t wins, ties to the lowest face index" as an order-independent minimum (t > 0: IEEE bit patterns
// order like the values; a face index is unique, so the lane whose key equals the slot afterwards is THE winner and records its
// leaf slot).  Same arithmetic per triangle as a per-lane loop over the leaf (triHitT_flat on the same operands), so the hit is bit-identical; a leaf
// step costs ~170 instructions per 64 pairs on full lanes instead of ~150 per two triangles per lane on the ~14 lanes that hold a
// leaf, for as many iterations as the fullest leaf needs (35 % of the walk's instructions, tools/trace_stats.py).
constexpr int COOP_PAIRS = 64 * 7;                            // most (ray, triangle) pairs of one step: 64 leaves of 7 triangles
struct CoopLeaf { unsigned* pairs; unsigned long long* best; int* slot; float2* rays; };   // this wave's LDS slices
// inclusive prefix sum over the 64 lanes on the VALU (DPP row shifts + row broadcasts; no LDS permutes)
__device__ __forceinline__ int wave_incl_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);     // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);     // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);     // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);     // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return x;
}
// Split walks (below): the result word of a ray is cl.best[the ray's OWNER lane] for as long as the wave walks -- the lanes at a
// leaf name their ray's owner in cl.slot and every tester reports straight to the owner's word.
__device__ __forceinline__ void coop_leaf_step(const uint4* tris, WalkRay& r, WalkStack& st, int& cur, const CoopLeaf& cl, int lane, int ray_owner) {
    const bool at_leaf = cur < 0 && cur != WALK_DONE;
    const int v = -cur - 1, first = v >> 3, cnt = at_leaf ? (v & 7) : 0;
    const int incl = wave_incl_scan(cnt);                      // inclusive prefix of the triangle counts over the wave
    const int total = __builtin_amdgcn_readlane(incl, 63), excl = incl - cnt;
    // (a ray's word starts as (bound, face index 0): the bound is a primitive's hit or FLT_MAX, and a face at exactly that distance
    // must NOT replace it (strict t_min > t in the reference's loop): its key (t, f >= 0) is never below (t, 0))
    if (at_leaf) {
        cl.slot[lane] = ray_owner;
        cl.rays[3 * lane] = make_float2(r.o.x, r.o.y);                        // the owners' rays: three 8-byte reads per tester
        cl.rays[3 * lane + 1] = make_float2(r.o.z, r.d.x);
        cl.rays[3 * lane + 2] = make_float2(r.d.y, r.d.z);
        for (int k = 0; k < cnt; k++) cl.pairs[excl + k] = ((unsigned)lane << 26) | (unsigned)(first + k);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int base = 0; base < total; base += 64) {             // (wave-uniform trip count)
        const int pi = base + lane;
        const bool work = pi < total;
        const unsigned pr = cl.pairs[work ? pi : 0];
        const int holder = (int)(pr >> 26), slot = (int)(pr & 0x3ffffffu);
        const float2 ra = cl.rays[3 * holder], rb = cl.rays[3 * holder + 1], rc = cl.rays[3 * holder + 2];
        const v3 o = V(ra.x, ra.y, rb.x), d = V(rb.y, rc.x, rc.y);
        const int target = cl.slot[holder];
        unsigned long long key = ~0ull;
        if (work) {
            const uint4* tr = tris + (unsigned)slot * 3u;
            const uint4 r0 = tr[0], r1 = tr[1], r2 = tr[2];
            const float ta = triHitT_flat(V(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z)),
                                          V(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y)),
                                          V(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x)), o, d);
            if (ta > 0.0f) {
                key = ((unsigned long long)__float_as_uint(ta) << 32) | (unsigned)r2.y;
                atomicMin(&cl.best[target], key);
            }
        }
        STAT_ADD(2, work ? 1 : 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (at_leaf) {
        const unsigned long long b = cl.best[ray_owner];       // the ray's word, whoever lowered it
        r.t_min = __uint_as_float((unsigned)(b >> 32));
        r.best_face = (int)(unsigned)b;
        STAT_ADD(4, 1);
        cur = st.empty() ? WALK_DONE : st.pop();
    }
    STAT_WAVE(3);
}

// ---- split walks [r5].  A wave that walks 64 rays to the end runs the UNION of their walks -- 47 node trips for 7 node visits per
// ray on the atrium -- because a few rays take 30-70 visits while the lanes of the others idle.  Here an idle lane takes over part
// of a busy lane's walk: the BOTTOM entry of its traversal stack (a whole subtree: the farthest child pushed at the shallowest
// node) together with a copy of its ray.  All parts of a ray's walk share one 64-bit LDS word, that of the ray's OWNER lane
// (cl.best[owner]): (bits of t, face index), lowered by atomicMin -- the reference's "first strictly smaller t wins, ties to the
// lowest face index" as an order-independent minimum (the word starts as (bound, 0): a face AT the bound -- a primitive's
// distance -- is never below it) -- and every part prunes with the word's distance, which only falls.  Which lane walks which
// subtree therefore changes how many boxes are visited, never the result: bit-exact with the unsplit walk and the brute-force
// loop.  The winner's leaf slot comes from its face index (TraceParams::fslot).
constexpr int STEAL_MIN_IDLE = 8;                              // idle lanes that make a round of takeovers worth its ~60 instructions (4 .. 16: +-0.5 %)
struct SplitWalk {
    int owner;                     // the lane whose ray this lane is walking (a part of)
    bool shared;                   // that ray is walked by more than one lane: refresh the pruning distance from its word
};
__device__ __forceinline__ void split_refresh(WalkRay& r, const SplitWalk& sw, const CoopLeaf& cl) {
    if (sw.shared) {                                           // the other parts of this ray's walk may have found something nearer
        const unsigned long long wb = cl.best[sw.owner];
        const float tb = __uint_as_float((unsigned)(wb >> 32));
        if (tb < r.t_min) { r.t_min = tb; r.best_face = (int)(unsigned)wb; }   // (a real face: the word only falls through faces)
    }
}
#ifndef AIPT_STEAL_GIVE
#define AIPT_STEAL_GIVE 4
#endif
constexpr int STEAL_GIVE = AIPT_STEAL_GIVE;                    // most entries one lane gives away per round (to as many takers)
__device__ __forceinline__ void steal_step(WalkRay& r, WalkStack& st, int& cur, SplitWalk& sw, const CoopLeaf& cl, int lane) {
    const unsigned long long idle = __ballot(cur == WALK_DONE);
    const int nidle = __popcll(idle);
    if (nidle < STEAL_MIN_IDLE) return;
    // entries [bot, bot + g) of a busy lane's stack, while they are in LDS
    int g = 0;
    if (cur != WALK_DONE) g = min(min(st.sp, STACK_LDS) - st.bot, STEAL_GIVE);
    g = max(g, 0);
    if (!__ballot(g > 0)) return;
    const int incl = wave_incl_scan(g), excl = incl - g;
    const int n = min(nidle, __builtin_amdgcn_readlane(incl, 63));
    const int taken = min(max(n - excl, 0), g);
    if (taken > 0) {
        for (int k = 0; k < taken; k++) cl.pairs[excl + k] = (unsigned)lane | ((unsigned)(st.bot + k) << 8) | ((unsigned)sw.owner << 16);
        cl.rays[3 * lane] = make_float2(r.o.x, r.o.y);
        cl.rays[3 * lane + 1] = make_float2(r.o.z, r.d.x);
        cl.rays[3 * lane + 2] = make_float2(r.d.y, r.d.z);
        st.bot += taken;
        if (st.sp == st.bot) { st.sp = 0; st.bot = 0; }        // (the entries themselves stay where they are until the takers have read them, below)
        sw.shared = true;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ri = __popcll(idle & ((1ull << lane) - 1ull));
    if (cur == WALK_DONE && ri < n) {
        const unsigned pr = cl.pairs[ri];
        const int dl = (int)(pr & 0xffu), b = (int)((pr >> 8) & 0xffu);
        sw.owner = (int)(pr >> 16);
        cur = st.lds[b * 256 + (dl - lane)];                   // the giver's stack column, entry b
        const float2 ra = cl.rays[3 * dl], rb = cl.rays[3 * dl + 1], rc = cl.rays[3 * dl + 2];
        const unsigned long long bound = cl.best[sw.owner];
        r.start(V(ra.x, ra.y, rb.x), V(rb.y, rc.x, rc.y), __uint_as_float((unsigned)(bound >> 32)));
        // the whole word, face index included: a face at the SAME distance as the ray's nearest so far wins with a lower index only
        // (index 0 of the untouched word "(bound, 0)": nothing at the bound's distance wins, as it must not)
        r.best_face = (int)(unsigned)bound;
        st.sp = 0; st.bot = 0;
        sw.shared = true;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void bvh4_nearest(const TraceParams& p, v3 o, v3 d, WalkStack& st, float& t_min, int& best_slot,
                                             const CoopLeaf& cl, int lane, bool walk) {
    const uint4* nodes = reinterpret_cast<const uint4*>(p.nodes);
    const uint4* tris = reinterpret_cast<const uint4*>(p.tris);
    WalkRay r;
    r.start(o, d, t_min);
    int cur = walk ? 0 : WALK_DONE;                            // (lanes whose ray misses the mesh box help: leaf steps, takeovers)
    st.sp = 0; st.bot = 0;
#ifdef AIPT_TRACE_STATS
    int my_visits = 0;
#define STAT_MINE() my_visits++
#else
#define STAT_MINE() do {} while (0)
#endif
    {
        // wave-level "while-while": every lane descends to its next leaf, then ALL 64 lanes share the leaves' triangle tests;
        // idle lanes take over parts of the busy lanes' walks (steal_step)
        const unsigned long long key0 = ((unsigned long long)__float_as_uint(t_min) << 32);
        SplitWalk sw{lane, false};
        cl.best[lane] = key0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        while (true) {
            while (true) {
                steal_step(r, st, cur, sw, cl, lane);
                if (!__ballot(cur >= 0)) break;
                if (cur >= 0) {
                    split_refresh(r, sw, cl);
                    STAT_ADD(0, 1); STAT_WAVE(1); STAT_MINE();
                    walk_node(nodes, r, st, cur);
                }
            }
            if (!__ballot(cur != WALK_DONE)) break;
            coop_leaf_step(tris, r, st, cur, cl, lane, sw.owner);
        }
        const unsigned long long res = cl.best[lane];          // (every leaf step ended behind a wave barrier)
        best_slot = -1;
        if (walk && res != key0) {                             // a face strictly below the bound (or, between faces, the lowest index)
            t_min = __uint_as_float((unsigned)(res >> 32));
            best_slot = p.fslot[(unsigned)res];
        }
    }
#if defined(AIPT_TRACE_STATS) && !defined(AIPT_TRACE_NO_COUNTERS)
    atomicMax(&g_trace_stats[5], (unsigned long long)my_visits);
    const int bucket = my_visits <= 4 ? 8 : my_visits <= 8 ? 9 : my_visits <= 16 ? 10 : my_visits <= 32 ? 11 : my_visits <= 64 ? 12 : my_visits <= 128 ? 13 : 14;
    atomicAdd(&g_trace_stats[bucket], 1ull);
#endif
}

// the full face record behind leaf slot `slot` (five 16-byte loads)
__device__ __forceinline__ void load_leaf_face(const TraceParams& p, int slot, DevFace& f) {
    const uint4* fp = reinterpret_cast<const uint4*>(p.lfaces + slot);
    const uint4 q0 = fp[0], q1 = fp[1], q2 = fp[2], q3 = fp[3], q4 = fp[4];
    uint32_t* fw = reinterpret_cast<uint32_t*>(&f);
    fw[0] = q0.x; fw[1] = q0.y; fw[2] = q0.z; fw[3] = q0.w; fw[4] = q1.x; fw[5] = q1.y; fw[6] = q1.z; fw[7] = q1.w;
    fw[8] = q2.x; fw[9] = q2.y; fw[10] = q2.z; fw[11] = q2.w; fw[12] = q3.x; fw[13] = q3.y; fw[14] = q3.z; fw[15] = q3.w;
    fw[16] = q4.x; fw[17] = q4.y; fw[18] = q4.z;
}

// generateRayFromCamera (pathtrace.cu:155-182) for pixel `pix` of frame `fr`
__device__ __forceinline__ void camera_ray(const TraceParams& p, const aipt_camera& cam, int pix, v3& o, v3& d) {
    const int x = pix % p.W, y = pix / p.W;
    const v3 view = V(cam.view[0], cam.view[1], cam.view[2]);
    const v3 right = V(cam.right[0], cam.right[1], cam.right[2]);
    const v3 up = V(cam.up[0], cam.up[1], cam.up[2]);
    o = V(cam.position[0], cam.position[1], cam.position[2]);
    float jx = 0.0f, jy = 0.0f;
    if (p.flags & AIPT_TRACE_AA) {
        uint32_t rng = make_seed(p.iter, pix, 0);    // uninitialised in the reference, defined as 0
        jx = u01(rng, -0.5f, 0.5f);
        jy = u01(rng, -0.5f, 0.5f);
    }
    float sx = (float)x - (float)cam.resolution[0] * 0.5f;
    float sy = (float)y - (float)cam.resolution[1] * 0.5f;
    if (p.flags & AIPT_TRACE_AA) { sx = sx + jx; sy = sy + jy; }
    d = vnormalize(vsub(vsub(view, vscale(vscale(right, cam.pixelLength[0]), sx)),
                        vscale(vscale(up, cam.pixelLength[1]), sy)));
}

// ---------------------------------------------------------------------------------------------- the bounce kernel
// MESH = false drops the triangle path (and its LDS traversal stack) from the instantiation used for primitive-only scenes.
#ifndef AIPT_TRACE_OCC
#define AIPT_TRACE_OCC 1
#endif
#ifndef AIPT_WALK_OCC
#define AIPT_WALK_OCC 1      // workgroups per CU (= waves per SIMD) the later-bounce mesh instantiation is compiled for
#endif
template <bool FIRST, bool MESH>
__global__ __launch_bounds__(256, (MESH && !FIRST) ? AIPT_WALK_OCC : AIPT_TRACE_OCC) void trace_bounce(const TraceParams p) {
    __shared__ int s_wave[4];
    // Cameras of the frames traced together.  They arrive as kernel arguments, are read from the kernel-argument segment with
    // scalar loads (wave-uniform index) and indexed per lane from this LDS copy: `p.cams[fr]` with a per-lane frame index would
    // compile to vector loads from the kernel-argument segment, 21 dwords per lane through L1 instead of one LDS read each.
    // (Round 2 suspected those vector loads of the co-residency corruption; round 3 found packed-fp32 instructions beside gapped
    // fp16 MFMAs to be the cause, DESIGN.md 5 -- the LDS copy stays as the cheaper access, not as a fence.)
    __shared__ aipt_camera s_cams[FIRST ? BMAX : 1];
    // dynamic LDS: [MESH: STACK_LDS x 256 stack words][primitives (<= MAXG_LDS)][materials (<= MAXM_LDS)], sized by the launch
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];
    int* s_stack = s_dyn;
    DevGeom* s_geoms = reinterpret_cast<DevGeom*>(s_dyn + (MESH ? STACK_LDS * 256 : 0));
    aipt_material* s_mats = reinterpret_cast<aipt_material*>(s_geoms + (p.ngeoms <= MAXG_LDS ? p.ngeoms : 0));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    PHASE_INIT();
    const int P = p.P;
    float4* S0 = p.st; float4* S1 = p.st + p.PS; float4* S2 = p.st + 2 * p.PS;
    // live paths entering the bounce (complete: the previous kernels on this stream have finished); this workgroup advances
    // the 256-path block vb of the live list
    const int n_in = FIRST ? p.PT : p.n_live[p.bounce];
    const int vb = blockIdx.x;
    if (!FIRST && vb * 256 >= n_in) {                    // whole workgroup beyond the list
        if (tid == 0) p.cnt[vb] = 0;
        if (p.nframes > 1 && tid < p.nframes) p.cntf[vb * BMAX + tid] = 0;
        return;
    }

    // ---- which path does this thread advance, and at which index would thrust::partition have left it?
    // Bounce 0: thread t = pixel t.  Later bounces walk the LIVE LIST written by trace_compact: entry t is the pixel of the
    // t-th live path in array order, so t is exactly the compacted array index that seeds the RNG (pathtrace.cu:351) --
    // every wave is full of live paths and the state planes are gathered/scattered through the pixel index.
    // primitives and materials into LDS: the candidate loop and the shader index them per lane
    const bool broad = p.ngeoms <= MAXG_LDS && !(p.flags & AIPT_TRACE_NO_BROAD_PHASE);
    const bool mats_lds = p.nmats <= MAXM_LDS && !(p.flags & 0x10000000u);     // (bit 28: debug, materials from HBM)
    if (broad) {
        const int nw = p.ngeoms * (int)(sizeof(DevGeom) / 4);
        for (int k = tid; k < nw; k += 256) reinterpret_cast<int*>(s_geoms)[k] = reinterpret_cast<const int*>(p.geoms)[k];
    }
    if (mats_lds) {
        const int nw = p.nmats * (int)(sizeof(aipt_material) / 4);
        for (int k = tid; k < nw; k += 256) reinterpret_cast<int*>(s_mats)[k] = reinterpret_cast<const int*>(p.mats)[k];
    }
    if (FIRST) {
        // all 256 threads copy the cameras (21 dwords each) from the kernel-argument segment: one or two loads per thread (one lane
        // per wave copying whole structs was a chain of ~100 dependent scalar loads and LDS stores in front of every workgroup)
        static_assert(sizeof(aipt_camera) % 4 == 0, "camera copy by dwords");
        const int nw = p.nframes * (int)(sizeof(aipt_camera) / 4);
        const int* src = reinterpret_cast<const int*>(p.cams);
        for (int k = tid; k < nw; k += 256) reinterpret_cast<int*>(s_cams)[k] = src[k];
    }
    __syncthreads();
    PHASE(0);

    const bool walk_mesh = MESH && !(FIRST && p.cache_mode == 2) && p.nfaces && !(p.flags & 0x40000000u);
    // LDS of the cooperative leaf step (coop_leaf_step): per wave, the (ray, triangle) pair list, the rays and the result slots
    __shared__ unsigned s_pairs[MESH ? 4 * COOP_PAIRS : 1];
    __shared__ unsigned long long s_best[MESH ? 256 : 1];
    __shared__ int s_slot[MESH ? 256 : 1];
    __shared__ float2 s_rays[MESH ? 768 : 1];
    const CoopLeaf cl{s_pairs + (MESH ? wave * COOP_PAIRS : 0), s_best + (MESH ? wave * 64 : 0),
                      s_slot + (MESH ? wave * 64 : 0), s_rays + (MESH ? wave * 192 : 0)};
    int i, idx, rem = 0;
    bool alive;
    const int t = vb * 256 + tid;
    if (FIRST) {
        i = t; idx = t; alive = t < p.PT; rem = p.trace_depth;
    } else {
        alive = t < n_in;
        i = alive ? p.live_in[t] : 0;
        idx = (p.flags & AIPT_TRACE_COMPACT) ? t : i;
    }
    // Batched trace: the frames are INTERLEAVED pixel by pixel -- path i = pixel * nframes + frame -- so that neighbouring
    // lanes hold the same pixel of consecutive frames.  The reference seeds its RNG with (iteration, index, depth) only
    // (the same numbers every frame), so under a slow camera pan those paths are near copies of each other: they
    // walk the same BVH nodes, and a wave's union of walks covers 64 / nframes distinct paths instead of 64.  The RNG index of
    // a path is its rank among the live paths of ITS frame (what an unbatched trace of that frame computes): bounce 0 the
    // pixel, later the per-frame rank trace_compact stored next to the live list.
    int fr = 0, pix = i;
    if (p.nframes > 1) {
        pix = i / p.nframes;
        fr = i - pix * p.nframes;
        if (FIRST || !(p.flags & AIPT_TRACE_COMPACT)) idx = pix;
        else idx = alive ? p.rank_in[t] : 0;
    }

    bool alive_after = false;
    // (per-path state lives across the two halves of the path's work: between them the whole wave, dead lanes included, meets in
    // the un-pooled BVH walk, whose leaf steps are shared by all 64 lanes)
    v3 o = V(0, 0, 0), d = V(1, 1, 1), col = V(0, 0, 0);
    float t_min = FLT_MAX;
    int materialid = -1;
    v3 hitP = V(0, 0, 0), normal = V(0, 0, 0);
    bool want_walk = false;
    if (alive) {
        if (FIRST) {                                                             // generateRayFromCamera :155-182
            camera_ray(p, s_cams[FIRST ? fr : 0], pix, o, d);
            col = V(1.0f, 1.0f, 1.0f);
        } else {
            const float4 a = S0[i], b = S1[i], c = S2[i];
            o = V(a.x, a.y, a.z);
            d = V(a.w, b.x, b.y);
            col = V(b.z, b.w, c.x);
            rem = __float_as_int(c.y);
        }

        // ---- computeIntersections :200-306 (primitives first, then the mesh; strict t_min > t keeps the first of equals)
        const bool from_cache = FIRST && p.cache_mode == 2;        // CACHE_BOUNCE, iter > 1 (pathtrace.cu:473-476)
        if (from_cache) {
            const float* c = p.cache + i;
            t_min = c[0]; materialid = __float_as_int(c[(size_t)P]);
            hitP = V(c[(size_t)2 * P], c[(size_t)3 * P], c[(size_t)4 * P]);
            normal = V(c[(size_t)5 * P], c[(size_t)6 * P], c[(size_t)7 * P]);
        } else if (p.flags & 0x20000000u) {
            t_min = 5.0f; materialid = 1; hitP = vadd(o, vscale(d, 5.0f)); normal = V(0, 1, 0);      // (timing ablation)
        } else if (broad) {
            // broad phase over all primitives (wave-uniform loop, scalar loads), then the exact tests on this lane's
            // candidates only, in index order (so "the first of equal distances wins" as in the reference's loop): a
            // wave runs max-over-lanes(candidates) exact tests instead of ngeoms
            // The exact test of the candidate whose padded box the ray enters FIRST runs first; every other candidate whose box
            // is entered clearly beyond that hit is skipped -- its exact test could only return a larger distance, which never
            // wins -- so a wave runs ~1 box and ~1 sphere test instead of max-over-lanes(candidates) of each (the exact tests
            // were 39 % of the bounce kernel's wave cycles on the mesh scene).  Equal distances resolve by index as in the loop.
            const v3 inv = V(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
            unsigned cand = 0;
            float near_tn = INFINITY;
            int near_g = -1;
            for (int gi = 0; gi < p.ngeoms; gi++) {
                float tn;
                if (maybe_hits(p.geoms[gi].lo, p.geoms[gi].hi, o, inv, tn)) {
                    cand |= 1u << gi;
                    if (tn < near_tn) { near_tn = tn; near_g = gi; }
                }
            }
            int best_g = -1;
            auto exact = [&](int gi) {
                const DevGeom& g = s_geoms[gi];
                v3 tp, tnn;
                float t = -1.0f;
                if (g.type == AIPT_GEOM_CUBE) t = boxTest(g, o, d, tp, tnn);
                else if (g.type == AIPT_GEOM_SPHERE) t = sphereTest(g, o, d, tp, tnn);
                if (t > 0.0f && (t_min > t || (t_min == t && gi < best_g))) {
                    t_min = t; materialid = g.materialid; hitP = tp; normal = tnn; best_g = gi;
                }
            };
            if (near_g >= 0) { exact(near_g); cand &= ~(1u << near_g); }
            while (cand) {
                const int gi = __builtin_ctz(cand);
                cand &= cand - 1;
                float tn;
                maybe_hits(s_geoms[gi].lo, s_geoms[gi].hi, o, inv, tn);
                if (!(tn > t_min * 1.0001f + 1e-4f)) exact(gi);           // NaN -> test
            }
        } else {
            for (int gi = 0; gi < p.ngeoms; gi++) {
                const DevGeom& g = p.geoms[gi];
                v3 tp, tn;
                float t = -1.0f;
                if (g.type == AIPT_GEOM_CUBE) t = boxTest(g, o, d, tp, tn);
                else if (g.type == AIPT_GEOM_SPHERE) t = sphereTest(g, o, d, tp, tn);
                if (t > 0.0f && t_min > t) { t_min = t; materialid = g.materialid; hitP = tp; normal = tn; }
            }
        }
        PHASE(1);
        if (walk_mesh && ((p.flags & AIPT_TRACE_NO_CULL) || rayAABB(o, d, p.box))) {   // RAY_CULLING true (:23, :258) / false (:270-281)
            if (p.flags & AIPT_TRACE_BRUTE_FORCE) {
                // the reference's loop: every face, in index order
                for (int fi = 0; fi < p.nfaces; fi++) {
                    v3 tp, tn;
                    const float t = triangleTest(p.faces[fi], o, d, tp, tn);
                    if (t > 0.0f && t_min > t) { t_min = t; materialid = p.faces[fi].materialid; hitP = tp; normal = tn; }
                }
            } else want_walk = true;
        }
    }
    if (MESH && __syncthreads_or(want_walk)) {           // (workgroup-uniform: the cooperative leaf step's LDS slices are per wave)
        int best_slot = -1;
        WalkStack st{s_stack + tid, p.stack_ovf + t, p.PS, 0, 0};
        bvh4_nearest(p, o, d, st, t_min, best_slot, cl, lane, want_walk);
        if (best_slot >= 0) {
            // the winning face, once: the reference's full test gives its hit point and shading normal (and the same t)
            DevFace f;
            load_leaf_face(p, best_slot, f);
            v3 tp, tn;
            const float tt = triangleTest(f, o, d, tp, tn);
            t_min = tt; materialid = f.materialid; hitP = tp; normal = tn;
        }
    }
    if (alive) {
        PHASE(2);
        const bool hit = materialid != -1;
        if (FIRST && p.cache_mode == 1) {                          // CACHE_BOUNCE, iter == 1 (:466-472)
            float* c = p.cache + i;
            c[0] = t_min; c[(size_t)P] = __int_as_float(materialid);
            c[(size_t)2 * P] = hitP.x; c[(size_t)3 * P] = hitP.y; c[(size_t)4 * P] = hitP.z;
            c[(size_t)5 * P] = normal.x; c[(size_t)6 * P] = normal.y; c[(size_t)7 * P] = normal.z;
        }
        if (p.sortkey) p.sortkey[t] = hit ? materialid : 0;        // the hit record's materialId (memset 0, :478)
        const v3 surfN = vnormalize(normal);
        const int x = pix % p.W, y = pix / p.W;
        const size_t gd = (size_t)fr * p.gbuf_frame + (size_t)y * p.stride + (size_t)(p.W - x - 1);   // h-flipped destination (:297-299)

        // ---- shadeMaterial :333-390
        int new_rem;
        if (hit) {
            uint32_t rng = make_seed(p.iter, idx, rem);
            const aipt_material m = mats_lds ? s_mats[materialid] : p.mats[materialid];
            if (m.emittance > 0.0f) {
                new_rem = 0;
                col = vscale(vmul(col, V(m.color[0], m.color[1], m.color[2])), m.emittance);
            } else {
                scatterRay(o, d, col, surfN, hitP, m, rng, p.flags);
                new_rem = rem - 1;
            }
        } else {
            col = V(0.0f, 0.0f, 0.0f);
            new_rem = 0;
        }
        if (FIRST && p.iter == 1) {
            // planes 3-9; a miss leaves zeros (the reference memsets dev_tensor in pathtraceInit every frame, F6)
            float* gb = p.gbuf + gd;
            gb[p.plane * 3] = hit ? normal.x : 0.0f;
            gb[p.plane * 4] = hit ? normal.y : 0.0f;
            gb[p.plane * 5] = hit ? normal.z : 0.0f;
            gb[p.plane * 6] = hit ? t_min : 0.0f;
            gb[p.plane * 7] = hit ? col.x : 0.0f;
            gb[p.plane * 8] = hit ? col.y : 0.0f;
            gb[p.plane * 9] = hit ? col.z : 0.0f;
            if (p.mat0) p.mat0[(size_t)fr * P + pix] = materialid;
        }
        if (new_rem == 0) {
            // finalGather + copy_data (:393-402, :81-94): image += colour; planes 0-2 = image / iter.  At iter 1 the image
            // starts from zero (pathtraceInit memsets it, :102), so it is just this path's colour.
            const float fiter = (float)p.iter;
            float* gb = p.gbuf + gd;
            const size_t ii = (size_t)fr * 3 * P + (size_t)y * p.W + (size_t)(p.W - x - 1);
            float ax = col.x, ay = col.y, az = col.z;
            if (p.iter > 1) { ax = p.image[ii] + col.x; ay = p.image[ii + P] + col.y; az = p.image[ii + 2 * (size_t)P] + col.z; }
            p.image[ii] = ax; p.image[ii + P] = ay; p.image[ii + 2 * (size_t)P] = az;
            gb[0] = ax / fiter;
            gb[p.plane] = ay / fiter;
            gb[p.plane * 2] = az / fiter;
        } else {
            S0[i] = make_float4(o.x, o.y, o.z, d.x);
            S1[i] = make_float4(d.y, d.z, col.x, col.y);
            S2[i] = make_float4(col.z, __int_as_float(new_rem), 0.0f, 0.0f);
            alive_after = true;
        }
        p.alive[t] = alive_after ? 1 : 0;
        PHASE(3);
    }

    // ---- live count of this 256-path block for the next bounce
    const unsigned long long m2 = __ballot(alive_after);
    if (lane == 0) s_wave[wave] = __popcll(m2);
    __syncthreads();
    if (tid == 0) {
        const int c = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
        p.cnt[vb] = c;
        if (c) atomicAdd(&p.n_live[p.bounce + 1], c);
    }
    if (p.nframes > 1) {                            // survivors per frame: this block's counts (for the per-frame ranks) and the totals
        __shared__ int s_cf[BMAX];
        if (tid < BMAX) s_cf[tid] = 0;
        __syncthreads();
        for (int f = 0; f < p.nframes; f++) {
            const int c = __popcll(__ballot(alive_after && fr == f));
            if (lane == 0 && c) atomicAdd(&s_cf[f], c);
        }
        __syncthreads();
        if (tid < p.nframes) {
            p.cntf[vb * BMAX + tid] = s_cf[tid];
            if (s_cf[tid]) atomicAdd(&p.n_live_f[(p.bounce + 1) * BMAX + tid], s_cf[tid]);
        }
    }
}

// Batched traces: exclusive prefix over the workgroups of the survivor counts, in place -- blockIdx 0: all frames together
// (cnt), blockIdx 1 + f: frame f on its own (cntf) -- so that trace_compact finds its bases with one load each instead of
// summing all lower workgroups (with nframes x more workgroups and nframes + 1 counters that sum would grow quadratically).
__global__ __launch_bounds__(1024) void trace_scan(const TraceParams p) {
    __shared__ int s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.bounce == 0 ? p.PT : p.n_live[p.bounce];
    const int nb = (n + 255) / 256;                            // workgroups of the bounce that hold paths
    const int f = (int)blockIdx.x - 1;
    int* a = f < 0 ? p.cnt : p.cntf + f;
    const int stride = f < 0 ? 1 : BMAX;
    int carry = 0;                                             // sum of everything before the current block of entries
    // SE consecutive entries per thread and round (their loads in flight together): a quarter of the rounds -- and of the barriers
    // and dependent round trips -- of one entry per thread (86 k entries per scan at 24 frames x 1280x720)
    constexpr int SE = 4;
    for (int k0 = 0; k0 < nb; k0 += 1024 * SE) {
        const int k = k0 + tid * SE;
        int c[SE];
#pragma unroll
        for (int e = 0; e < SE; e++) c[e] = k + e < nb ? a[(size_t)(k + e) * stride] : 0;
        int tsum = 0;
#pragma unroll
        for (int e = 0; e < SE; e++) tsum += c[e];
        int incl = tsum;                                       // inclusive scan of the threads' sums inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < 16; w++) { const int v = s_wsum[w]; if (w < wave) woff += v; total += v; }
        int run = carry + woff + incl - tsum;                  // exclusive prefix of this thread's first entry
#pragma unroll
        for (int e = 0; e < SE; e++) {
            if (k + e < nb) a[(size_t)(k + e) * stride] = run;
            run += c[e];
        }
        carry += total;
        __syncthreads();
    }
}

// Stable stream compaction of the live list (the wave64 ballot/popcount analogue of thrust::partition, pathtrace.cu:505):
// array slot t of the bounce that just ran survives iff its path still has bounces left; survivors keep their order.
// Output position = (live counts of the lower workgroups, written by trace_bounce) + (ballot prefix inside the workgroup).
__global__ __launch_bounds__(256) void trace_compact(const TraceParams p) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.bounce == 0 ? p.PT : p.n_live[p.bounce];
    if ((int)(blockIdx.x * 256) >= n) return;
    const int t = blockIdx.x * 256 + tid;
    int i = 0;
    bool alive = false;
    if (t < n) {
        i = p.live_in ? p.live_in[t] : t;
        alive = p.alive[t] != 0;
    }
    int part = 0;
    if (p.nframes > 1) part = lane == 0 && wave == 0 ? p.cnt[blockIdx.x] : 0;      // batched: trace_scan left the exclusive prefix
    else {
        for (int j = tid; j < (int)blockIdx.x; j += 256) part += p.cnt[j];
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    }
    const unsigned long long mask = __ballot(alive);
    if (lane == 0) s_wave[wave] = part;
    __syncthreads();
    if (tid == 0) s_base = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
    __syncthreads();
    const int base = s_base;
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(mask);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += s_wave[w];
    const int slot = base + woff + __popcll(mask & ((1ull << lane) - 1ull));
    if (alive) p.live_out[slot] = i;
    if (p.nframes > 1) {
        // per-frame rank of every survivor = survivors of ITS frame in lower workgroups + in lower waves + in lower lanes
        __shared__ int s_fbase[BMAX], s_fw[4][BMAX];
        const int fr = i % p.nframes;
        if (tid < BMAX) s_fbase[tid] = tid < p.nframes ? p.cntf[blockIdx.x * BMAX + tid] : 0;   // exclusive prefix (trace_scan)
        int my_rank = 0;
        for (int f = 0; f < p.nframes; f++) {
            const unsigned long long mf = __ballot(alive && fr == f);
            if (lane == 0) s_fw[wave][f] = __popcll(mf);
            if (fr == f) my_rank = __popcll(mf & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (alive) {
            int r = s_fbase[fr] + my_rank;
            for (int w = 0; w < wave; w++) r += s_fw[w][fr];
            p.rank_out[slot] = r;
        }
    }
}

// ---- AIPT_TRACE_SORT_MATERIAL: stable counting sort of the compacted live list by material id ------------------------------
// The reference sorts the surviving PathSegments with thrust::sort_by_key (pathtrace.cu:508-510; a stable merge sort).  Its
// keys are the hit records the PRE-partition array slots hold, so slot j of the compacted list is keyed by sortkey[j], the
// material id the bounce kernel found for array slot j (0 for a miss).  Three small launches over the 4-byte list:
// per-workgroup histograms, one exclusive scan in (key, workgroup) order, stable scatter by ballot rank.
__global__ __launch_bounds__(256) void trace_sort_hist(const TraceParams p) {
    __shared__ int s_cnt[256];
    const int tid = threadIdx.x;
    const int n = p.n_live[p.bounce + 1];
    s_cnt[tid] = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + tid;
    if (t < n) atomicAdd(&s_cnt[p.sortkey[t]], 1);
    __syncthreads();
    if (tid < p.nkeys) p.hist[tid * p.nblk + blockIdx.x] = s_cnt[tid];
}
__global__ __launch_bounds__(1024) void trace_sort_scan(const TraceParams p) {
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    const int total = p.nkeys * p.nblk;
    const int per = (total + 1023) / 1024;
    const int lo = tid * per, hi = min(total, lo + per);
    int sum = 0;
    for (int k = lo; k < hi; k++) sum += p.hist[k];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                       // inclusive Hillis-Steele scan of the 1024 partial sums
        const int v = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int k = lo; k < hi; k++) { const int c = p.hist[k]; p.hist[k] = run; run += c; }
}
__global__ __launch_bounds__(256) void trace_sort_scatter(const TraceParams p) {
    __shared__ int s_wave[4][256];                              // per-wave count of every key
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n_live[p.bounce + 1];
    if ((int)(blockIdx.x * 256) >= n) return;
    const int t = blockIdx.x * 256 + tid;
    const bool in = t < n;
    const int key = in ? p.sortkey[t] : -1;
    int rank = 0;
    for (int k = 0; k < p.nkeys; k++) {                        // wave-uniform loop; keys are few (materials)
        const unsigned long long m = __ballot(key == k);
        if (key == k) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave][k] = __popcll(m);
    }
    __syncthreads();
    if (in) {
        int off = p.hist[key * p.nblk + blockIdx.x];
        for (int w = 0; w < wave; w++) off += s_wave[w][key];
        p.sort_out[off + rank] = p.sort_in[t];
    }
}

static void free_scene(TraceState* s) {
    hipFree(s->d_geoms); hipFree(s->d_mats); hipFree(s->d_faces); hipFree(s->d_nodes); hipFree(s->d_tris); hipFree(s->d_lfaces); hipFree(s->d_fslot);
    s->d_geoms = nullptr; s->d_mats = nullptr; s->d_faces = nullptr; s->have_scene = false;
    s->d_nodes = nullptr; s->d_tris = nullptr; s->d_lfaces = nullptr; s->d_fslot = nullptr; s->nnodes = 0; s->cache_valid = false;
}
static void free_frame(TraceState* s) {
    hipFree(s->d_state); hipFree(s->d_cnt); hipFree(s->d_alive); hipFree(s->d_nlive); hipFree(s->d_mat0); hipFree(s->d_image);
    for (int*& l : s->d_live) { hipFree(l); l = nullptr; }
    hipFree(s->d_cache); hipFree(s->d_sortkey); hipFree(s->d_hist); hipFree(s->d_stack_ovf);
    hipFree(s->d_cams); hipFree(s->d_nlive_f); hipFree(s->d_cntf); hipFree(s->d_rank[0]); hipFree(s->d_rank[1]);
    s->d_cams = nullptr; s->d_nlive_f = nullptr; s->d_cntf = nullptr; s->d_rank[0] = s->d_rank[1] = nullptr;
    s->d_stack_ovf = nullptr; s->ovf_entries = 0;
    s->d_state = nullptr; s->d_cnt = nullptr; s->d_alive = nullptr; s->d_nlive = nullptr; s->d_mat0 = nullptr; s->d_image = nullptr;
    s->d_cache = nullptr; s->d_sortkey = nullptr; s->d_hist = nullptr; s->hist_keys = 0; s->cache_valid = false;
}
static void free_trace_profile(TraceState* s) {
    for (hipEvent_t e : s->prof_ev) if (e) hipEventDestroy(e);
    s->prof_ev.clear();
    s->prof_frames.clear();
    s->prof_max = 0; s->prof_calls = 0; s->prof_seen = 0;
}

static void free_side(TraceState* s) {
    if (!s->side) return;
    free_frame(s->side);                        // (the scene pointers of a side lane are borrowed)
    delete s->side;
    s->side = nullptr;
}

void trace_destroy(aipt_ctx* ctx) {
    TraceState* s = ctx->trace;
    if (!s) return;
    free_side(s);
    free_scene(s);
    free_frame(s);
    free_trace_profile(s);
    delete s;
    ctx->trace = nullptr;
}

static TraceState* tstate(aipt_ctx* ctx) {
    if (!ctx->trace) ctx->trace = new TraceState();
    return ctx->trace;
}

// device form of one primitive: the three matrices + the padded world box of the broad phase
static DevGeom make_dev_geom(const aipt_geom& g) {
    DevGeom d;
    d.type = g.type; d.materialid = g.materialid;
    memcpy(d.inv, g.inverseTransform, 64);
    memcpy(d.xf, g.transform, 64);
    memcpy(d.invT, g.invTranspose, 64);
    // padded world box of the unit cube [-0.5, 0.5]^3 under the primitive's transform (encloses the unit sphere too)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    const float* m = g.transform;                             // column-major
    for (int c = 0; c < 8; c++) {
        const double x = (c & 1) ? 0.5 : -0.5, y = (c & 2) ? 0.5 : -0.5, z = (c & 4) ? 0.5 : -0.5;
        for (int r = 0; r < 3; r++) {
            const double v = (double)m[0 * 4 + r] * x + (double)m[1 * 4 + r] * y + (double)m[2 * 4 + r] * z + (double)m[3 * 4 + r];
            if (v < lo[r]) lo[r] = v;
            if (v > hi[r]) hi[r] = v;
        }
    }
    double scale = 1.0;
    for (int r = 0; r < 3; r++) { scale = fmax(scale, hi[r] - lo[r]); scale = fmax(scale, fmax(fabs(lo[r]), fabs(hi[r]))); }
    for (int r = 0; r < 3; r++) {
        d.lo[r] = (float)(lo[r] - 1e-3 * scale);
        d.hi[r] = (float)(hi[r] + 1e-3 * scale);
        if (!(d.lo[r] <= d.hi[r])) { d.lo[r] = -3.0e38f; d.hi[r] = 3.0e38f; }   // NaN/inf transform: never cull
    }
    return d;
}

// ---- packed scene: everything a context needs to trace, relocatable, built once (rank 0) and broadcast to the other GPUs.
// Layout: PackHeader, then 16-byte aligned sections in this order: geoms (aipt_geom), materials (aipt_material), faces
// (aipt_face, caller's order), BVH nodes (Bvh4Node), leaf triangle records (TriRec; TriRec::face = index into faces).
struct PackHeader {
    char magic[8];                 // "AIPTSB02"
    uint32_t ngeoms, nmats, nfaces, nnodes;
    int32_t stack_need;
    uint32_t reserved[3];
    aipt_aabb box;
    uint64_t total_bytes;
};
static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
struct PackView {
    const PackHeader* h;
    const aipt_geom* geoms; const aipt_material* mats; const aipt_face* faces; const Bvh4Node* nodes; const TriRec* tris;
};
static size_t pack_layout(uint32_t ng, uint32_t nm, uint32_t nf, uint32_t nn, size_t off[5]) {
    size_t o = align16(sizeof(PackHeader));
    off[0] = o; o = align16(o + sizeof(aipt_geom) * (size_t)ng);
    off[1] = o; o = align16(o + sizeof(aipt_material) * (size_t)nm);
    off[2] = o; o = align16(o + sizeof(aipt_face) * (size_t)nf);
    off[3] = o; o = align16(o + sizeof(Bvh4Node) * (size_t)nn);
    off[4] = o; o = align16(o + sizeof(TriRec) * (size_t)nf);
    return o;
}
static bool pack_view(const void* blob, size_t bytes, PackView& v) {
    if (!blob || bytes < sizeof(PackHeader)) return false;
    const PackHeader* h = (const PackHeader*)blob;
    if (memcmp(h->magic, "AIPTSB02", 8)) return false;
    size_t off[5];
    const size_t total = pack_layout(h->ngeoms, h->nmats, h->nfaces, h->nnodes, off);
    if (total != bytes || h->total_bytes != bytes) return false;
    const char* b = (const char*)blob;
    v.h = h;
    v.geoms = (const aipt_geom*)(b + off[0]); v.mats = (const aipt_material*)(b + off[1]); v.faces = (const aipt_face*)(b + off[2]);
    v.nodes = (const Bvh4Node*)(b + off[3]); v.tris = (const TriRec*)(b + off[4]);
    return true;
}

}  // namespace aipt

using namespace aipt;

extern "C" {

int aipt_scene_pack(const aipt_geom* geoms, int ngeoms, const aipt_material* materials, int nmaterials,
                    const aipt_face* faces, int nfaces, const aipt_aabb* mesh_box, void** blob_out, size_t* bytes_out,
                    char* err, size_t errlen) {
    auto bad = [&](int code, const char* fmt, int a, int b, int c) {
        if (err && errlen) snprintf(err, errlen, fmt, a, b, c);
        return code;
    };
    if (!blob_out || !bytes_out) return AIPT_E_INVALID;
    *blob_out = nullptr; *bytes_out = 0;
    if (ngeoms < 0 || nmaterials <= 0 || nfaces < 0 || (ngeoms && !geoms) || !materials || (nfaces && (!faces || !mesh_box)))
        return bad(AIPT_E_INVALID, "aipt_scene_pack: bad arguments (%d geoms, %d materials, %d faces)", ngeoms, nmaterials, nfaces);
    for (int i = 0; i < ngeoms; i++)
        if (geoms[i].materialid < 0 || geoms[i].materialid >= nmaterials)
            return bad(AIPT_E_INVALID, "geom %d: material %d of %d", i, geoms[i].materialid, nmaterials);
    for (int i = 0; i < nfaces; i++)
        if (faces[i].materialid < 0 || faces[i].materialid >= nmaterials)
            return bad(AIPT_E_INVALID, "face %d: material %d of %d", i, faces[i].materialid, nmaterials);
    std::vector<Bvh4Node> nodes;
    std::vector<int> lidx;
    int need = 0;
    if (nfaces) {
        need = build_bvh4(faces, nfaces, nodes, lidx);
        if (need < 0) return bad(AIPT_E_INVALID, "aipt_scene_pack: mesh of %d faces needs a deeper traversal stack than %d entries%c", nfaces, BVH_MAX_STACK, ' ');
    }
    size_t off[5];
    const size_t total = pack_layout(ngeoms, nmaterials, nfaces, (uint32_t)nodes.size(), off);
    char* b = (char*)calloc(1, total);
    if (!b) return bad(AIPT_E_NOMEM, "aipt_scene_pack: out of memory (%d geoms, %d materials, %d faces)", ngeoms, nmaterials, nfaces);
    PackHeader* h = (PackHeader*)b;
    memcpy(h->magic, "AIPTSB02", 8);
    h->ngeoms = ngeoms; h->nmats = nmaterials; h->nfaces = nfaces; h->nnodes = (uint32_t)nodes.size();
    h->stack_need = need;
    if (nfaces) h->box = *mesh_box;
    h->total_bytes = total;
    if (ngeoms) memcpy(b + off[0], geoms, sizeof(aipt_geom) * (size_t)ngeoms);
    memcpy(b + off[1], materials, sizeof(aipt_material) * (size_t)nmaterials);
    if (nfaces) {
        memcpy(b + off[2], faces, sizeof(aipt_face) * (size_t)nfaces);
        memcpy(b + off[3], nodes.data(), sizeof(Bvh4Node) * nodes.size());
        TriRec* tr = (TriRec*)(b + off[4]);
        for (int k = 0; k < nfaces; k++) {
            const aipt_face& f = faces[lidx[k]];
            for (int a = 0; a < 3; a++) {
                tr[k].v0[a] = f.v[0][a];
                tr[k].e1[a] = f.v[1][a] - f.v[0][a];         // intersect.inl:44-45, the same fp32 subtraction
                tr[k].e2[a] = f.v[2][a] - f.v[0][a];
            }
            tr[k].face = lidx[k];
        }
    }
    *blob_out = b; *bytes_out = total;
    return AIPT_OK;
}

void aipt_blob_free(void* blob) { free(blob); }

int aipt_scene_upload_packed(aipt_ctx* ctx, const void* blob, size_t bytes) {
    AIPT_CHECK_CTX(ctx);
    PackView v;
    if (!pack_view(blob, bytes, v)) return fail(ctx, AIPT_E_FORMAT, "aipt_scene_upload_packed: not a packed scene (%zu bytes)", bytes);
    const int ngeoms = (int)v.h->ngeoms, nmats = (int)v.h->nmats, nfaces = (int)v.h->nfaces, nnodes = (int)v.h->nnodes;
    if (nmats <= 0 || (nfaces && (!nnodes || v.h->stack_need < 0 || v.h->stack_need > BVH_MAX_STACK)))
        return fail(ctx, AIPT_E_FORMAT, "aipt_scene_upload_packed: inconsistent header");
    if (nfaces >= (1 << 26))       // the cooperative leaf step packs (lane, leaf slot) into 32 bits; leaf references hold slot * 8 + count
        return fail(ctx, AIPT_E_INVALID, "aipt_scene_upload_packed: %d faces (at most %d)", nfaces, (1 << 26) - 1);
    for (int i = 0; i < ngeoms; i++)
        if (v.geoms[i].materialid < 0 || v.geoms[i].materialid >= nmats) return fail(ctx, AIPT_E_FORMAT, "packed scene: geom %d material", i);
    for (int k = 0; k < nfaces; k++)
        if (v.tris[k].face < 0 || v.tris[k].face >= nfaces || v.faces[k].materialid < 0 || v.faces[k].materialid >= nmats)
            return fail(ctx, AIPT_E_FORMAT, "packed scene: face record %d", k);
    // leaf slot of every face (the walk's winner is a face index): every face in exactly one leaf record -- checked here, with the
    // other record checks, BEFORE the loaded scene is touched (a malformed blob leaves the context as it was)
    std::vector<int> fslot(nfaces, -1);
    for (int k = 0; k < nfaces; k++) {
        if (fslot[v.tris[k].face] >= 0) return fail(ctx, AIPT_E_FORMAT, "packed scene: face %d is in two leaf records", v.tris[k].face);
        fslot[v.tris[k].face] = k;
    }
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    TraceState* s = tstate(ctx);
    free_scene(s);
    std::vector<DevGeom> dg(ngeoms);
    for (int i = 0; i < ngeoms; i++) dg[i] = make_dev_geom(v.geoms[i]);
    s->h_geoms.assign(v.geoms, v.geoms + ngeoms);
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_geoms, sizeof(DevGeom) * (ngeoms ? ngeoms : 1)));
    if (ngeoms) AIPT_HIP(ctx, hipMemcpy(s->d_geoms, dg.data(), sizeof(DevGeom) * ngeoms, hipMemcpyHostToDevice));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_mats, sizeof(aipt_material) * nmats));
    AIPT_HIP(ctx, hipMemcpy(s->d_mats, v.mats, sizeof(aipt_material) * nmats, hipMemcpyHostToDevice));
    static_assert(sizeof(DevFace) == sizeof(aipt_face), "face layout");
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_faces, sizeof(DevFace) * (nfaces ? nfaces : 1)));
    if (nfaces) {
        AIPT_HIP(ctx, hipMemcpy(s->d_faces, v.faces, sizeof(DevFace) * nfaces, hipMemcpyHostToDevice));
        s->box = v.h->box;
        std::vector<DevFaceP> lf(nfaces);
        for (int k = 0; k < nfaces; k++) { memcpy(&lf[k].f, &v.faces[v.tris[k].face], sizeof(DevFace)); lf[k].pad = 0; }
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_nodes, sizeof(Bvh4Node) * nnodes));
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_tris, sizeof(TriRec) * nfaces));
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_lfaces, sizeof(DevFaceP) * nfaces));
        AIPT_HIP(ctx, hipMemcpy(s->d_nodes, v.nodes, sizeof(Bvh4Node) * nnodes, hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMemcpy(s->d_tris, v.tris, sizeof(TriRec) * nfaces, hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMemcpy(s->d_lfaces, lf.data(), sizeof(DevFaceP) * nfaces, hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_fslot, sizeof(int) * nfaces));
        AIPT_HIP(ctx, hipMemcpy(s->d_fslot, fslot.data(), sizeof(int) * nfaces, hipMemcpyHostToDevice));
        s->nnodes = nnodes;
    } else {
        memset(&s->box, 0, sizeof(s->box));
    }
    s->stack_need = v.h->stack_need;
    s->ngeoms = ngeoms; s->nmats = nmats; s->nfaces = nfaces;
    s->have_scene = true; s->cache_valid = false;
    return AIPT_OK;
}

int aipt_scene_upload(aipt_ctx* ctx, const aipt_geom* geoms, int ngeoms, const aipt_material* materials, int nmaterials,
                      const aipt_face* faces, int nfaces, const aipt_aabb* mesh_box) {
    AIPT_CHECK_CTX(ctx);
    void* blob = nullptr;
    size_t bytes = 0;
    char err[256] = "";
    int rc = aipt_scene_pack(geoms, ngeoms, materials, nmaterials, faces, nfaces, mesh_box, &blob, &bytes, err, sizeof(err));
    if (rc) return fail(ctx, rc, "%s", err[0] ? err : "aipt_scene_upload: bad arguments");
    rc = aipt_scene_upload_packed(ctx, blob, bytes);
    aipt_blob_free(blob);
    return rc;
}

int aipt_scene_free(aipt_ctx* ctx) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    free_scene(tstate(ctx));
    return AIPT_OK;
}

__global__ void trace_scan_warm(int* p) { if (threadIdx.x == 1000) p[0] = 0; }     // first submission to the side lane's stream

static int configure_state(aipt_ctx* ctx, TraceState* s, int width, int height, int batch) {
    if (s->W == width && s->H == height && s->batch == batch && s->d_state) return AIPT_OK;
    free_frame(s);
    const int P = width * height;
    const size_t PT = (size_t)P * batch;
    const int nblk = (int)((PT + 255) / 256);
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_state, sizeof(float4) * 3 * PT));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_cnt, sizeof(int) * (PT / 64 + 16)));       // a later bounce can run 64-path workgroups
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_alive, sizeof(int) * PT));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_nlive, sizeof(int) * (MAX_DEPTH + 1)));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_nlive_f, sizeof(int) * (MAX_DEPTH + 1) * BMAX));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_cams, sizeof(aipt_camera) * BMAX));
    if (batch > 1) {
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_cntf, sizeof(int) * (size_t)nblk * BMAX));
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_rank[0], sizeof(int) * PT));
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_rank[1], sizeof(int) * PT));
    }
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_mat0, sizeof(int) * PT));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_image, sizeof(float) * 3 * PT));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_live[0], sizeof(int) * PT));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_live[1], sizeof(int) * PT));
    s->W = width; s->H = height; s->P = P; s->nblk = nblk; s->batch = batch;
    s->mat0_valid = false; s->cache_valid = false;
    return AIPT_OK;
}

int aipt_trace_configure_batch(aipt_ctx* ctx, int width, int height, int batch) {
    AIPT_CHECK_CTX(ctx);
    if (width <= 0 || height <= 0 || batch < 1 || batch > BMAX || (long)width * height * batch > (1l << 30))
        return fail(ctx, AIPT_E_INVALID, "aipt_trace_configure: %dx%d x %d frames (1..%d)", width, height, batch, BMAX);
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    TraceState* s = tstate(ctx);
    if (s->side && (s->W != width || s->H != height || s->batch != batch)) free_side(s);
    return configure_state(ctx, s, width, height, batch);
}

extern "C++" { namespace aipt {
// aipt_frames_configure: the second lane of aipt_frames' two-lane traces, allocated here and not inside somebody's timed frame
int trace_enable_lanes(aipt_ctx* ctx) {
    TraceState* s = tstate(ctx);
    if (!s->d_state || s->batch < AIPT_TRACE_LANES_MIN) return AIPT_OK;
    // (AIPT_TRACE_LANES=1, the profiling passes: no second lane, so none of its buffers -- half a batch of path state)
    static const int lanes_env = getenv("AIPT_TRACE_LANES") ? atoi(getenv("AIPT_TRACE_LANES")) : 2;
    if (lanes_env < 2) { free_side(s); return AIPT_OK; }
    if (!s->side) { s->side = new TraceState(); s->side->is_side = true; }
    const int rc2 = configure_state(ctx, s->side, s->W, s->H, (s->batch + 1) / 2);
    if (rc2) {
        // no memory for the second lane is not an error of aipt_frames_configure: a call is then traced by one lane
        // (trace_lanes_ready() is false without it), as before round 5
        (void)hipGetLastError();
        free_side(s);
        ctx->err.clear();
        return AIPT_OK;
    }
    // ... together with its stream, which is made to run something now: the first submission to a new HIP stream sets up its
    // hardware queue (milliseconds -- measured inside a timed 20-frame call: 715 instead of 905 frames/s)
    if (!ctx->st_lane1) {
        AIPT_HIP(ctx, hipStreamCreateWithFlags(&ctx->st_lane1, hipStreamNonBlocking));
        AIPT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_lane_fork, hipEventDisableTiming));
        AIPT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_lane_join, hipEventDisableTiming));
    }
    AIPT_HIP(ctx, hipMemsetAsync(s->side->d_nlive, 0, sizeof(int) * (MAX_DEPTH + 1), ctx->st_lane1));
    hipLaunchKernelGGL(trace_scan_warm, dim3(1), dim3(64), 0, ctx->st_lane1, s->side->d_nlive);
    AIPT_HIP(ctx, hipEventRecord(ctx->ev_lane_join, ctx->st_lane1));
    AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_lane_join, 0));
    AIPT_HIP(ctx, hipStreamSynchronize(ctx->st_lane1));
    return AIPT_OK;
}
} }

int aipt_trace_configure(aipt_ctx* ctx, int width, int height) { return aipt_trace_configure_batch(ctx, width, height, 1); }

int aipt_trace_batch(aipt_ctx* ctx, const aipt_camera* cams, int nframes, int iter, int depth, uint32_t flags,
                     float* d_gbuf, int gbuf_rows, int gbuf_stride, size_t gbuf_frame_floats) {
    AIPT_CHECK_CTX(ctx);
    return aipt::trace_on_stream(ctx, ctx->stream, cams, nframes, iter, depth, flags, d_gbuf, gbuf_rows, gbuf_stride, gbuf_frame_floats);
}

int aipt_trace(aipt_ctx* ctx, const aipt_camera* cam, int iter, int depth, uint32_t flags,
               float* d_gbuf, int gbuf_rows, int gbuf_stride) {
    AIPT_CHECK_CTX(ctx);
    return aipt::trace_on_stream(ctx, ctx->stream, cam, 1, iter, depth, flags, d_gbuf, gbuf_rows, gbuf_stride, 0);
}

extern "C++" {
namespace aipt {
int trace_on_stream(aipt_ctx* ctx, hipStream_t st, const aipt_camera* cam, int nframes, int iter, int depth, uint32_t flags,
                    float* d_gbuf, int gbuf_rows, int gbuf_stride, size_t gbuf_frame, int lane) {
    TraceState* s = tstate(ctx);
    if (!s->have_scene) return fail(ctx, AIPT_E_STATE, "aipt_trace: no scene uploaded");
    if (!s->d_state) return fail(ctx, AIPT_E_STATE, "aipt_trace: call aipt_trace_configure first");
    if (lane) {
        // lane 1: its own path state (half of the batch), the main lane's scene.  Always on the SAME side stream, which orders
        // its traces among themselves; the caller orders it against everything else (trace_frames in abi.cpp)
        TraceState* m = s;
        TraceState* d = m->side;
        if (!d || !d->d_state || d->W != m->W || d->H != m->H) return fail(ctx, AIPT_E_STATE, "aipt_trace: the side lane is not configured");
        d->d_geoms = m->d_geoms; d->ngeoms = m->ngeoms; d->d_mats = m->d_mats; d->nmats = m->nmats;
        d->d_faces = m->d_faces; d->nfaces = m->nfaces; d->stack_need = m->stack_need;
        d->d_nodes = m->d_nodes; d->nnodes = m->nnodes; d->d_tris = m->d_tris; d->d_lfaces = m->d_lfaces; d->d_fslot = m->d_fslot;
        d->box = m->box; d->have_scene = true;
        s = d;
    }
    if (!cam || !d_gbuf) return fail(ctx, AIPT_E_INVALID, "aipt_trace: NULL argument");
    if (nframes < 1 || nframes > s->batch)
        return fail(ctx, AIPT_E_INVALID, "aipt_trace: %d frames, configured for %d (aipt_trace_configure_batch)", nframes, s->batch);
    for (int f = 0; f < nframes; f++)
        if (cam[f].resolution[0] != s->W || cam[f].resolution[1] != s->H)
            return fail(ctx, AIPT_E_INVALID, "aipt_trace: camera is %dx%d, configured %dx%d", cam[f].resolution[0],
                        cam[f].resolution[1], s->W, s->H);
    if (nframes > 1 && (iter != 1 || (flags & (AIPT_TRACE_SORT_MATERIAL | AIPT_TRACE_CACHE_FIRST_BOUNCE | AIPT_TRACE_MOTION_BLUR))))
        return fail(ctx, AIPT_E_INVALID, "aipt_trace_batch: a batch holds iteration-1 frames without the sort / cache / motion-blur toggles");
    if (nframes > 1 && gbuf_frame < (size_t)10 * gbuf_rows * gbuf_stride)
        return fail(ctx, AIPT_E_INVALID, "aipt_trace_batch: G-buffers of consecutive frames overlap");
    if (depth < 1 || depth > MAX_DEPTH) return fail(ctx, AIPT_E_INVALID, "aipt_trace: depth %d not in 1..%d", depth, MAX_DEPTH);
    if (iter < 1) return fail(ctx, AIPT_E_INVALID, "aipt_trace: iter %d (iterations count from 1, main.cpp:149)", iter);
    if (gbuf_rows < s->H || gbuf_stride < s->W) return fail(ctx, AIPT_E_INVALID, "aipt_trace: G-buffer %dx%d too small", gbuf_rows, gbuf_stride);
    const bool sortmat = (flags & AIPT_TRACE_SORT_MATERIAL) != 0, cache = (flags & AIPT_TRACE_CACHE_FIRST_BOUNCE) != 0;
    const bool blur = (flags & AIPT_TRACE_MOTION_BLUR) != 0;
    if (sortmat && !(flags & AIPT_TRACE_COMPACT))
        return fail(ctx, AIPT_E_INVALID, "aipt_trace: AIPT_TRACE_SORT_MATERIAL needs AIPT_TRACE_COMPACT");
    if (sortmat && s->nmats > 256) return fail(ctx, AIPT_E_INVALID, "aipt_trace: AIPT_TRACE_SORT_MATERIAL supports up to 256 materials (%d)", s->nmats);
    if (cache && ((flags & AIPT_TRACE_AA) || blur))             // the reference's asserts, pathtrace.cu:435-436
        return fail(ctx, AIPT_E_INVALID, "aipt_trace: AIPT_TRACE_CACHE_FIRST_BOUNCE is only legal without AIPT_TRACE_AA and AIPT_TRACE_MOTION_BLUR");
    if (cache && s->batch > 1) return fail(ctx, AIPT_E_INVALID, "aipt_trace: AIPT_TRACE_CACHE_FIRST_BOUNCE needs a single-frame configuration");
    if (cache && !s->d_cache) AIPT_HIP(ctx, hipMalloc((void**)&s->d_cache, sizeof(float) * 8 * (size_t)s->P));
    if (cache && iter > 1 && !s->cache_valid)
        return fail(ctx, AIPT_E_STATE, "aipt_trace: AIPT_TRACE_CACHE_FIRST_BOUNCE at iter %d without a cached iter 1", iter);
    if (sortmat) {
        if (!s->d_live[2]) AIPT_HIP(ctx, hipMalloc((void**)&s->d_live[2], sizeof(int) * (size_t)s->P * s->batch));
        if (!s->d_sortkey) AIPT_HIP(ctx, hipMalloc((void**)&s->d_sortkey, sizeof(int) * (size_t)s->P * s->batch));
        if (s->hist_keys < s->nmats) {
            AIPT_HIP(ctx, hipStreamSynchronize(st));
            hipFree(s->d_hist); s->d_hist = nullptr;
            AIPT_HIP(ctx, hipMalloc((void**)&s->d_hist, sizeof(int) * (size_t)s->nmats * s->nblk));
            s->hist_keys = s->nmats;
        }
    }
    const int ovf_need = s->nfaces && s->stack_need > STACK_LDS ? s->stack_need - STACK_LDS : 0;
    // (every trace of the main lane also sizes the side lane's overflow area, so that a call's first two-lane trace does not
    // allocate -- and synchronise -- inside somebody's timed frames)
    for (TraceState* t : {s, !lane && s->side && s->side->d_state ? s->side : (TraceState*)nullptr}) {
        if (!t || ovf_need <= t->ovf_entries) continue;
        AIPT_HIP(ctx, hipStreamSynchronize(st));
        if (t != s && ctx->st_lane1) AIPT_HIP(ctx, hipStreamSynchronize(ctx->st_lane1));
        hipFree(t->d_stack_ovf); t->d_stack_ovf = nullptr; t->ovf_entries = 0;
        AIPT_HIP(ctx, hipMalloc((void**)&t->d_stack_ovf, sizeof(int) * (size_t)ovf_need * t->P * t->batch));
        t->ovf_entries = ovf_need;
    }
    if (!lane && ctx->last_trace_stream && ctx->last_trace_stream != st) AIPT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_traced, 0));
    if (blur && !(iter % 4) && iter < 3000) {                   // moveGeom, pathtrace.cu:442-446 (dt = 0.10)
        bool moved = false;
        for (aipt_geom& g : s->h_geoms) {
            if (g.vel[0] == 0.0f && g.vel[1] == 0.0f && g.vel[2] == 0.0f) continue;
            for (int a = 0; a < 3; a++) g.translation[a] += g.vel[a] * 0.10f;
            aipt_geom_build(&g);
            moved = true;
        }
        if (moved) {
            std::vector<DevGeom> dg(s->ngeoms);
            for (int i = 0; i < s->ngeoms; i++) dg[i] = make_dev_geom(s->h_geoms[i]);
            AIPT_HIP(ctx, hipStreamSynchronize(st));            // earlier traces still read the old primitives
            AIPT_HIP(ctx, hipMemcpy(s->d_geoms, dg.data(), sizeof(DevGeom) * s->ngeoms, hipMemcpyHostToDevice));
        }
    }
    TraceParams p;
    for (int f = 0; f < nframes; f++) p.cams[f] = cam[f];
    p.iter = iter; p.trace_depth = depth; p.flags = flags;
    p.W = s->W; p.H = s->H; p.P = s->P;
    p.nframes = nframes; p.PT = nframes * s->P; p.PS = (size_t)s->P * s->batch; p.gbuf_frame = gbuf_frame;
    p.n_live_f = s->d_nlive_f;
    const int nblk = (p.PT + 255) / 256;
    p.cntf = s->d_cntf; p.rank_in = nullptr; p.rank_out = nullptr;
    if (nframes > 1) AIPT_HIP(ctx, hipMemsetAsync(s->d_nlive_f, 0, sizeof(int) * (MAX_DEPTH + 1) * BMAX, st));
    p.st = s->d_state;
    p.geoms = s->d_geoms; p.ngeoms = s->ngeoms; p.mats = s->d_mats; p.nmats = s->nmats;
    p.faces = s->d_faces; p.nfaces = s->nfaces; p.box = s->box;
    p.nodes = s->d_nodes; p.tris = s->d_tris; p.lfaces = s->d_lfaces; p.fslot = s->d_fslot;
    p.gbuf = d_gbuf; p.plane = (size_t)gbuf_rows * gbuf_stride; p.stride = gbuf_stride;
    p.cnt = s->d_cnt; p.alive = s->d_alive;
    p.n_live = s->d_nlive;
    p.mat0 = (flags & AIPT_TRACE_RECORD_MAT0) ? s->d_mat0 : nullptr;
    p.image = s->d_image;
    p.cache = s->d_cache; p.cache_mode = cache ? (iter == 1 ? 1 : 2) : 0;
    p.sortkey = sortmat ? s->d_sortkey : nullptr;
    p.hist = s->d_hist; p.nkeys = s->nmats; p.nblk = nblk;
    p.sort_in = nullptr; p.sort_out = nullptr;
    p.stack_ovf = s->d_stack_ovf;
    const bool prof = s->prof_max && s->prof_calls < s->prof_max && s->prof_seen % s->prof_every == 0;
    AIPT_HIP(ctx, hipMemsetAsync(s->d_nlive, 0, sizeof(int) * (MAX_DEPTH + 1), st));
    const bool mesh = s->nfaces > 0;
    const size_t lds_scene = (s->ngeoms <= MAXG_LDS ? sizeof(DevGeom) * s->ngeoms : 0) + (s->nmats <= MAXM_LDS ? sizeof(aipt_material) * s->nmats : 0);
    const size_t stack_bytes = (mesh ? (size_t)STACK_LDS * 256 * sizeof(int) : 0) + lds_scene;     // dynamic LDS of the bounce kernels
    snprintf(s->kname[0], sizeof(s->kname[0]), "trace_bounce<true,%s>", mesh ? "true" : "false");
    snprintf(s->kname[1], sizeof(s->kname[1]), "trace_bounce<false,%s>", mesh ? "true" : "false");
    int cur = -1;                                               // live list the bounce reads (-1: bounce 0, all pixels)
    for (int b = 0; b < depth; b++) {
        p.bounce = b;
        p.live_in = cur < 0 ? nullptr : s->d_live[cur];
        const int nxt = cur < 0 ? 0 : (cur + 1) % (sortmat ? 3 : 2);
        p.live_out = s->d_live[nxt];
        if (nframes > 1) { p.rank_in = cur < 0 ? nullptr : s->d_rank[cur]; p.rank_out = s->d_rank[nxt]; }
        hipEvent_t* pev = prof ? &s->prof_ev[((size_t)s->prof_calls * MAX_DEPTH + b) * 2] : nullptr;
        if (pev) AIPT_HIP(ctx, hipEventRecord(pev[0], st));
        if (b == 0 && mesh) hipLaunchKernelGGL((trace_bounce<true, true>), dim3(nblk), dim3(256), stack_bytes, st, p);
        else if (b == 0) hipLaunchKernelGGL((trace_bounce<true, false>), dim3(nblk), dim3(256), lds_scene, st, p);
        else if (mesh) hipLaunchKernelGGL((trace_bounce<false, true>), dim3(nblk), dim3(256), stack_bytes, st, p);
        else hipLaunchKernelGGL((trace_bounce<false, false>), dim3(nblk), dim3(256), lds_scene, st, p);
        if (pev) AIPT_HIP(ctx, hipEventRecord(pev[1], st));
        if (b + 1 < depth) {
            if (nframes > 1) hipLaunchKernelGGL(trace_scan, dim3(1 + nframes), dim3(1024), 0, st, p);
            hipLaunchKernelGGL(trace_compact, dim3(nblk), dim3(256), 0, st, p);
            cur = nxt;
            if (sortmat) {                                      // compacted list -> sorted list (thrust::sort_by_key, :508-510)
                const int srt = (cur + 1) % 3;
                p.sort_in = s->d_live[cur]; p.sort_out = s->d_live[srt];
                hipLaunchKernelGGL(trace_sort_hist, dim3(nblk), dim3(256), 0, st, p);
                hipLaunchKernelGGL(trace_sort_scan, dim3(1), dim3(1024), 0, st, p);
                hipLaunchKernelGGL(trace_sort_scatter, dim3(nblk), dim3(256), 0, st, p);
                cur = srt;
            }
        }
    }
    if (cache && iter == 1) s->cache_valid = true;
    if (s->prof_max) {
        if (prof) { s->prof_calls++; s->prof_frames.push_back(nframes); }
        s->prof_seen++;
    }
    AIPT_HIP(ctx, hipGetLastError());
    if (!lane) {
        AIPT_HIP(ctx, hipEventRecord(ctx->ev_traced, st));
        ctx->last_trace_stream = st;
    }
    if (lane) tstate(ctx)->side_used = true; else s->side_used = false;
    s->last_depth = depth;
    s->last_frames = nframes;
    s->mat0_valid = p.mat0 != nullptr;
    return AIPT_OK;
}
}  // namespace aipt
}  // extern "C++"

int aipt_trace_profile_begin(aipt_ctx* ctx, int max_calls, int every) {
    AIPT_CHECK_CTX(ctx);
    if (max_calls < 1 || max_calls > 4096 || every < 1) return fail(ctx, AIPT_E_INVALID, "aipt_trace_profile_begin: max_calls %d, every %d", max_calls, every);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    TraceState* s = tstate(ctx);
    free_trace_profile(s);
    s->prof_ev.resize((size_t)max_calls * MAX_DEPTH * 2, nullptr);
    for (hipEvent_t& e : s->prof_ev) AIPT_HIP(ctx, hipEventCreate(&e));
    s->prof_max = max_calls; s->prof_every = every;
    return AIPT_OK;
}

int aipt_trace_profile_end(aipt_ctx* ctx, double* sum_ms_per_bounce, int nbounces, int* calls) {
    AIPT_CHECK_CTX(ctx);
    TraceState* s = tstate(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    if (sum_ms_per_bounce) {
        for (int b = 0; b < nbounces; b++) sum_ms_per_bounce[b] = 0.0;
        for (int c = 0; c < s->prof_calls; c++)
            for (int b = 0; b < nbounces && b < s->last_depth && b < MAX_DEPTH; b++) {
                float ms = 0;
                AIPT_HIP(ctx, hipEventElapsedTime(&ms, s->prof_ev[((size_t)c * MAX_DEPTH + b) * 2], s->prof_ev[((size_t)c * MAX_DEPTH + b) * 2 + 1]));
                sum_ms_per_bounce[b] += ms;
            }
    }
    if (calls) *calls = s->prof_calls;
    free_trace_profile(s);
    return AIPT_OK;
}

int aipt_trace_profile_calls(aipt_ctx* ctx, int* nframes_per_call, double* ms_per_call_bounce, int nbounces, int max_calls, int* calls) {
    AIPT_CHECK_CTX(ctx);
    TraceState* s = tstate(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    const int n = s->prof_calls < max_calls ? s->prof_calls : max_calls;
    for (int c = 0; c < n; c++) {
        if (nframes_per_call) nframes_per_call[c] = s->prof_frames[c];
        for (int b = 0; ms_per_call_bounce && b < nbounces; b++) {
            float ms = 0;
            if (b < s->last_depth && b < MAX_DEPTH)
                AIPT_HIP(ctx, hipEventElapsedTime(&ms, s->prof_ev[((size_t)c * MAX_DEPTH + b) * 2], s->prof_ev[((size_t)c * MAX_DEPTH + b) * 2 + 1]));
            ms_per_call_bounce[(size_t)c * nbounces + b] = ms;
        }
    }
    if (calls) *calls = n;
    return AIPT_OK;
}

extern "C++" { namespace aipt {
bool trace_profiling(aipt_ctx* ctx) { return ctx->trace && ctx->trace->prof_max > 0; }
bool trace_lanes_ready(aipt_ctx* ctx, int nframes) {
    const TraceState* s = ctx->trace;
    return s && s->side && s->side->d_state && s->side->W == s->W && s->side->H == s->H && nframes <= s->side->batch && ctx->st_lane1;
}
} }

int aipt_trace_kernel_name(aipt_ctx* ctx, int bounce, char* kernel, size_t kernel_len) {
    AIPT_CHECK_CTX(ctx);
    TraceState* s = tstate(ctx);
    if (!s->last_depth) return fail(ctx, AIPT_E_STATE, "aipt_trace_kernel_name: no trace has run");
    if (!kernel || !kernel_len) return fail(ctx, AIPT_E_INVALID, "aipt_trace_kernel_name: NULL buffer");
    strncpy(kernel, s->kname[bounce > 0 ? 1 : 0], kernel_len - 1);
    kernel[kernel_len - 1] = 0;
    return AIPT_OK;
}

int aipt_debug_trace_stats(aipt_ctx* ctx, unsigned long long* out8, int reset) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
#ifdef AIPT_TRACE_STATS
    if (out8) AIPT_HIP(ctx, hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_trace_stats), 128));
    if (out8 && reset == 2) AIPT_HIP(ctx, hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase), 64));     // reset == 2: the phase cycle sums instead
    if (reset == 1) { unsigned long long z8[8] = {0}; AIPT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z8, 64)); }
    if (reset == 1) { unsigned long long z[16] = {0}; AIPT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_trace_stats), z, 128)); }
    return AIPT_OK;
#else
    (void)reset;
    if (out8) memset(out8, 0, 128);
    return fail(ctx, AIPT_E_STATE, "aipt_debug_trace_stats: library built without -DAIPT_TRACE_STATS");
#endif
}

int aipt_trace_live_counts(aipt_ctx* ctx, int* h_n_live, int n) {
    AIPT_CHECK_CTX(ctx);
    TraceState* s = tstate(ctx);
    if (!s->d_nlive || !s->last_depth) return fail(ctx, AIPT_E_STATE, "aipt_trace_live_counts: no trace has run");
    if (!h_n_live || n < 1) return fail(ctx, AIPT_E_INVALID, "aipt_trace_live_counts: bad arguments");
    std::vector<int> tmp(MAX_DEPTH + 1);
    if (ctx->last_trace_stream) AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_traced, 0));
    AIPT_HIP(ctx, hipMemcpyAsync(tmp.data(), s->d_nlive, sizeof(int) * (MAX_DEPTH + 1), hipMemcpyDeviceToHost, ctx->stream));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    tmp[0] = s->P * s->last_frames;
    // a two-lane call (aipt_frames): the second half of its frames was traced on the side lane -- the totals cover both halves
    if (s->side_used && s->side && s->side->d_nlive) {
        std::vector<int> side(MAX_DEPTH + 1);
        AIPT_HIP(ctx, hipMemcpy(side.data(), s->side->d_nlive, sizeof(int) * (MAX_DEPTH + 1), hipMemcpyDeviceToHost));
        side[0] = s->side->P * s->side->last_frames;
        for (int i = 0; i <= s->last_depth; i++) tmp[i] += side[i];
    }
    for (int i = 0; i < n; i++) h_n_live[i] = i <= s->last_depth ? tmp[i] : 0;
    return AIPT_OK;
}

int aipt_trace_live_counts_frame(aipt_ctx* ctx, int frame, int* h_n_live, int n) {
    AIPT_CHECK_CTX(ctx);
    TraceState* s = tstate(ctx);
    if (!s->d_nlive || !s->last_depth) return fail(ctx, AIPT_E_STATE, "aipt_trace_live_counts_frame: no trace has run");
    // a two-lane call (aipt_frames): its second half was traced on the side lane; frames count through both halves
    if (s->side_used && s->side && frame >= s->last_frames) { frame -= s->last_frames; s = s->side; }
    if (!h_n_live || n < 1 || frame < 0 || frame >= s->last_frames) return fail(ctx, AIPT_E_INVALID, "aipt_trace_live_counts_frame: bad arguments");
    if (s->last_frames == 1 && !s->is_side) return aipt_trace_live_counts(ctx, h_n_live, n);
    if (s->last_frames == 1) {                     // a one-frame side lane keeps totals only
        std::vector<int> one(MAX_DEPTH + 1);
        AIPT_HIP(ctx, aipt::sync_streams(ctx));
        AIPT_HIP(ctx, hipMemcpy(one.data(), s->d_nlive, sizeof(int) * (MAX_DEPTH + 1), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) h_n_live[i] = i == 0 ? s->P : (i <= s->last_depth ? one[i] : 0);
        return AIPT_OK;
    }
    std::vector<int> tmp((MAX_DEPTH + 1) * BMAX);
    if (ctx->last_trace_stream) AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_traced, 0));
    AIPT_HIP(ctx, hipMemcpyAsync(tmp.data(), s->d_nlive_f, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost, ctx->stream));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    for (int i = 0; i < n; i++) h_n_live[i] = i == 0 ? s->P : (i <= s->last_depth ? tmp[i * BMAX + frame] : 0);
    return AIPT_OK;
}

int aipt_trace_first_hit_materials(aipt_ctx* ctx, int* h_mat, int n) {
    AIPT_CHECK_CTX(ctx);
    TraceState* s = tstate(ctx);
    if (!s->mat0_valid) return fail(ctx, AIPT_E_STATE, "aipt_trace_first_hit_materials: last trace did not record them");
    // a two-lane call (aipt_frames): the main lane's frames, then the side lane's (each lane in its own batch layout: path = pixel x
    // the lane's frame count + frame within the lane)
    const TraceState* const side = s->side_used && s->side && s->side->mat0_valid && s->side->d_mat0 ? s->side : nullptr;
    const int n_main = s->P * s->last_frames, n_side = side ? side->P * side->last_frames : 0;
    if (!h_mat || n != n_main + n_side) return fail(ctx, AIPT_E_INVALID, "aipt_trace_first_hit_materials: n=%d, expected %d", n, n_main + n_side);
    if (ctx->last_trace_stream) AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_traced, 0));
    AIPT_HIP(ctx, hipMemcpyAsync(h_mat, s->d_mat0, sizeof(int) * (size_t)n_main, hipMemcpyDeviceToHost, ctx->stream));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    if (side) AIPT_HIP(ctx, hipMemcpy(h_mat + n_main, side->d_mat0, sizeof(int) * (size_t)n_side, hipMemcpyDeviceToHost));
    return AIPT_OK;
}

}  // extern "C"
