// trace.hip -- 1-spp wavefront path tracer for CDNA4 (gfx950), hand-written HIP.
//
// Replaces pathtraceInit / pathtrace / pathtraceFree (reference Inference/src/pathtrace.h:6-8, pathtrace.cu:96-145,
// 422-528) and the device math of intersections.h / interactions.h.  Results are defined by the reference's kernels
// (generateRayFromCamera :155-182, computeIntersections :200-306, shadeMaterial :333-390, thrust::partition :505,
// finalGather :393-402, copy_data :81-94); the execution plan is not:
//
//  * Path state is SoA in HBM (9 float planes + 1 int plane, 40 B/path) and NEVER MOVES: path i is pixel i for the
//    whole frame, so every access is a fully coalesced wave64 load/store.
//  * Stream compaction is logical.  thrust::partition keeps survivors in their original relative order, so the index
//    a live path has after compaction -- which seeds its RNG, pathtrace.cu:351 -- is its RANK among live paths in pixel
//    order.  Each bounce kernel recomputes that rank from per-workgroup live counts written by the previous bounce
//    (block prefix) + wave64 __ballot/__popcll (in-block prefix).  No scan kernel, no scatter pass, no host sync per
//    bounce (the reference syncs twice per bounce, :483/:505).
//  * One kernel per bounce fuses ray generation (bounce 0), nearest-hit search, shading/scatter, the G-buffer
//    writes (planes 3-9, already h-flipped) and finalGather+copy_data: a path deposits its colour into planes 0-2
//    the moment its remainingBounces reaches 0 (light, miss, or depth exhausted -- SURVEY F9).
//
// Arithmetic is plain IEEE fp32 (this file is compiled with -ffp-contract=off, correctly rounded divide/sqrt), in the
// statement order of the reference/GLM sources, so the output matches the CPU restatement used by the tests bit for
// bit.  sin/cos of the hemisphere angle use the same Cody-Waite + minimax polynomial on both sides (det_sincosf).
#include "internal.h"

#include <cfloat>
#include <cstring>

namespace aipt {

struct v3 { float x, y, z; };

struct DevGeom {          // one primitive; matrices column-major (glm::mat4)
    int type, materialid;
    float inv[16], xf[16], invT[16];
    float lo[3], hi[3];   // padded world-space box around the primitive (broad phase only, never decides a hit)
};
constexpr int MAXG_LDS = 32;   // scenes with up to this many primitives use the per-lane candidate loop
struct DevFace {          // Face, sceneStructs.h:40
    float v[3][3], n[3][3];
    int materialid;
};

struct TraceState {
    DevGeom* d_geoms = nullptr; int ngeoms = 0;
    aipt_material* d_mats = nullptr; int nmats = 0;
    DevFace* d_faces = nullptr; int nfaces = 0;
    int bvh_depth = 0;
    BvhNode* d_nodes = nullptr; int nnodes = 0;     // threaded BVH over the faces (bvh.cpp)
    DevFace* d_lfaces = nullptr;                    // faces in leaf order
    int* d_lidx = nullptr;                          // their original indices (tie-break + parity with the index-ordered loop)
    aipt_aabb box{};
    bool have_scene = false;
    int W = 0, H = 0, P = 0, nblk = 0;
    float* d_state = nullptr;     // [10][P]: ox oy oz dx dy dz cr cg cb rem(int bits)
    int* d_cnt[2] = {nullptr, nullptr};   // per-workgroup live counts, ping-pong between bounces
    int* d_live[2] = {nullptr, nullptr};  // live-path index lists, ping-pong between bounces
    int* d_nlive = nullptr;       // [MAX_DEPTH+1]
    int* d_mat0 = nullptr;        // [P]
    float* d_image = nullptr;     // [3][P] radiance accumulated over iterations 1..n (dev_image, pathtrace.cu:101), h-flipped
    float* d_cache = nullptr;     // [8][P] bounce-0 hit records of iteration 1 (AIPT_TRACE_CACHE_FIRST_BOUNCE): t, material, P, raw N
    int* d_live3 = nullptr;       // third live list (AIPT_TRACE_SORT_MATERIAL: compact -> sort -> next bounce)
    int* d_sortkey = nullptr;     // [P] material id of the hit found in array slot t at the current bounce (0 = miss)
    int* d_hist = nullptr;        // [nkeys][nblk] counting-sort histogram / bases
    int hist_keys = 0;
    bool cache_valid = false;
    std::vector<aipt_geom> h_geoms;   // host copy for AIPT_TRACE_MOTION_BLUR (moveGeom, pathtrace.cu:318-331)
    int last_depth = 0;
    bool mat0_valid = false;
};
constexpr int MAX_DEPTH = 64;

struct TraceParams {
    aipt_camera cam;
    int iter, trace_depth, bounce;
    uint32_t flags;
    int W, H, P;
    float* st;
    const DevGeom* geoms; int ngeoms;
    const aipt_material* mats;
    const DevFace* faces; int nfaces;
    const BvhNode* nodes; int nnodes;
    const DevFace* lfaces; const int* lidx;
    aipt_aabb box;
    float* gbuf; size_t plane; int stride;
    const int* cnt_in; int* cnt_out;
    int* n_live;
    int* mat0;
    float* image;
    const int* live_in;          // pixel indices of the live paths entering this bounce, in array order (nullptr: all pixels)
    int* live_out;
    float* cache; int cache_mode;   // 0: none, 1: save the bounce-0 hit records, 2: reuse them instead of intersecting
    int* sortkey;                   // AIPT_TRACE_SORT_MATERIAL: material id of the hit in array slot t (0 = miss)
    int* hist; int nkeys, nblk;
    const int* sort_in; int* sort_out;
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
__device__ __forceinline__ bool maybe_hits(const float* lo, const float* hi, v3 o, v3 inv) {
    const float t1 = (lo[0] - o.x) * inv.x, t2 = (hi[0] - o.x) * inv.x;
    const float t3 = (lo[1] - o.y) * inv.y, t4 = (hi[1] - o.y) * inv.y;
    const float t5 = (lo[2] - o.z) * inv.z, t6 = (hi[2] - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
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

// scatterRay, live branch (DIELECTRIC false, FRESNELS true): :194-258
__device__ void scatterRay(v3& origin, v3& direction, v3& pcolor, v3 hitN, v3 hitP, const aipt_material& m, uint32_t& rng) {
    v3 dir = direction;
    v3 color;
    const v3 mcolor = V(m.color[0], m.color[1], m.color[2]);
    const v3 scolor = V(m.specular_color[0], m.specular_color[1], m.specular_color[2]);
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
    pcolor = vmul(pcolor, color);
}

// ---------------------------------------------------------------------------------------------- the bounce kernel
// MESH = false drops the triangle path (and its LDS traversal stack) from the instantiation used for primitive-only scenes.
template <bool FIRST, bool MESH>
__global__ __launch_bounds__(256) void trace_bounce(const TraceParams p) {
    __shared__ int s_wave[4];
    __shared__ DevGeom s_geoms[MAXG_LDS];
    extern __shared__ int s_stack[];                            // MESH: [tree depth + 1][thread] far children still to visit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = p.P;
    float* ox = p.st;            float* oy = p.st + (size_t)P;      float* oz = p.st + (size_t)2 * P;
    float* dx = p.st + (size_t)3 * P; float* dy = p.st + (size_t)4 * P; float* dz = p.st + (size_t)5 * P;
    float* cr = p.st + (size_t)6 * P; float* cg = p.st + (size_t)7 * P; float* cb = p.st + (size_t)8 * P;
    int* remp = reinterpret_cast<int*>(p.st + (size_t)9 * P);

    // ---- which path does this thread advance, and at which index would thrust::partition have left it?
    // Bounce 0: thread t = pixel t.  Later bounces walk the LIVE LIST written by trace_compact: entry t is the pixel of the
    // t-th live path in pixel order, so t is exactly the compacted array index that seeds the RNG (pathtrace.cu:351) --
    // every wave is full of live paths and the state planes are gathered/scattered through the (monotonic) pixel index.
    int i, idx, rem = 0;
    bool alive;
    const int t = blockIdx.x * 256 + tid;
    if (FIRST) {
        i = t; idx = t; alive = t < P; rem = p.trace_depth;
    } else {
        const int n = p.n_live[p.bounce];              // complete: the previous kernels on this stream have finished
        if ((int)(blockIdx.x * 256) >= n) {            // whole workgroup beyond the list
            if (tid == 0) p.cnt_out[blockIdx.x] = 0;
            return;
        }
        alive = t < n;
        i = alive ? p.live_in[t] : 0;
        idx = (p.flags & AIPT_TRACE_COMPACT) ? t : i;
        if (alive) rem = remp[i];
    }

    // primitives into LDS: the candidate loop below indexes them per lane
    const bool broad = p.ngeoms <= MAXG_LDS && !(p.flags & AIPT_TRACE_NO_BROAD_PHASE);
    if (broad) {
        const int nw = p.ngeoms * (int)(sizeof(DevGeom) / 4);
        for (int k = tid; k < nw; k += 256) reinterpret_cast<int*>(s_geoms)[k] = reinterpret_cast<const int*>(p.geoms)[k];
        __syncthreads();
    }

    bool alive_after = false;
    if (alive) {
        v3 o, d, col;
        if (FIRST) {                                                             // generateRayFromCamera :155-182
            const int x = i % p.W, y = i / p.W;
            const v3 view = V(p.cam.view[0], p.cam.view[1], p.cam.view[2]);
            const v3 right = V(p.cam.right[0], p.cam.right[1], p.cam.right[2]);
            const v3 up = V(p.cam.up[0], p.cam.up[1], p.cam.up[2]);
            o = V(p.cam.position[0], p.cam.position[1], p.cam.position[2]);
            col = V(1.0f, 1.0f, 1.0f);
            float jx = 0.0f, jy = 0.0f;
            if (p.flags & AIPT_TRACE_AA) {
                uint32_t rng = make_seed(p.iter, i, 0);      // uninitialised in the reference, defined as 0
                jx = u01(rng, -0.5f, 0.5f);
                jy = u01(rng, -0.5f, 0.5f);
            }
            float sx = (float)x - (float)p.cam.resolution[0] * 0.5f;
            float sy = (float)y - (float)p.cam.resolution[1] * 0.5f;
            if (p.flags & AIPT_TRACE_AA) { sx = sx + jx; sy = sy + jy; }
            d = vnormalize(vsub(vsub(view, vscale(vscale(right, p.cam.pixelLength[0]), sx)),
                                vscale(vscale(up, p.cam.pixelLength[1]), sy)));
        } else {
            o = V(ox[i], oy[i], oz[i]);
            d = V(dx[i], dy[i], dz[i]);
            col = V(cr[i], cg[i], cb[i]);
        }

        // ---- computeIntersections :200-306 (primitives first, then the mesh; strict t_min > t keeps the first of equals)
        float t_min = FLT_MAX;
        int materialid = -1;
        v3 hitP = V(0, 0, 0), normal = V(0, 0, 0);
        const bool from_cache = FIRST && p.cache_mode == 2;        // CACHE_BOUNCE, iter > 1 (pathtrace.cu:473-476)
        if (from_cache) {
            const float* c = p.cache + i;
            t_min = c[0]; materialid = __float_as_int(c[(size_t)P]);
            hitP = V(c[(size_t)2 * P], c[(size_t)3 * P], c[(size_t)4 * P]);
            normal = V(c[(size_t)5 * P], c[(size_t)6 * P], c[(size_t)7 * P]);
        } else
        if (broad) {
            // broad phase over all primitives (wave-uniform loop, scalar loads), then the exact tests on this lane's
            // candidates only, in index order (so "the first of equal distances wins" as in the reference's loop): a
            // wave runs max-over-lanes(candidates) exact tests instead of ngeoms
            const v3 inv = V(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
            unsigned cand = 0;
            for (int gi = 0; gi < p.ngeoms; gi++)
                if (maybe_hits(p.geoms[gi].lo, p.geoms[gi].hi, o, inv)) cand |= 1u << gi;
            while (cand) {
                const int gi = __builtin_ctz(cand);
                cand &= cand - 1;
                const DevGeom& g = s_geoms[gi];
                v3 tp, tn;
                float t = -1.0f;
                if (g.type == AIPT_GEOM_CUBE) t = boxTest(g, o, d, tp, tn);
                else if (g.type == AIPT_GEOM_SPHERE) t = sphereTest(g, o, d, tp, tn);
                if (t > 0.0f && t_min > t) { t_min = t; materialid = g.materialid; hitP = tp; normal = tn; }
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
        if (MESH && !from_cache && p.nfaces && rayAABB(o, d, p.box)) {                   // RAY_CULLING true (:23, :258)
            if (p.flags & AIPT_TRACE_BRUTE_FORCE) {
                // the reference's loop: every face, in index order
                for (int fi = 0; fi < p.nfaces; fi++) {
                    v3 tp, tn;
                    const float t = triangleTest(p.faces[fi], o, d, tp, tn);
                    if (t > 0.0f && t_min > t) { t_min = t; materialid = p.faces[fi].materialid; hitP = tp; normal = tn; }
                }
            } else {
                // BVH, front to back: same triangle test on the candidate faces.  The index-ordered loop keeps the FIRST face
                // among equal distances and never lets a face replace a primitive at equal distance; best_face
                // reproduces exactly that, so the result is the brute-force result.
                const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
                int best_face = -1;
                int sp = 0, cur = 0;                           // reference being visited: >= 0 inner node, < 0 leaf
                while (true) {
                    if (cur >= 0) {
                        const BvhNode nd = p.nodes[cur];
                        const float a1 = (nd.lo0[0] - o.x) * ix, a2 = (nd.hi0[0] - o.x) * ix;
                        const float a3 = (nd.lo0[1] - o.y) * iy, a4 = (nd.hi0[1] - o.y) * iy;
                        const float a5 = (nd.lo0[2] - o.z) * iz, a6 = (nd.hi0[2] - o.z) * iz;
                        const float b1 = (nd.lo1[0] - o.x) * ix, b2 = (nd.hi1[0] - o.x) * ix;
                        const float b3 = (nd.lo1[1] - o.y) * iy, b4 = (nd.hi1[1] - o.y) * iy;
                        const float b5 = (nd.lo1[2] - o.z) * iz, b6 = (nd.hi1[2] - o.z) * iz;
                        const float n0 = fmaxf(fmaxf(fminf(a1, a2), fminf(a3, a4)), fminf(a5, a6));
                        const float f0 = fminf(fminf(fmaxf(a1, a2), fmaxf(a3, a4)), fmaxf(a5, a6));
                        const float n1 = fmaxf(fmaxf(fminf(b1, b2), fminf(b3, b4)), fminf(b5, b6));
                        const float f1 = fminf(fminf(fmaxf(b1, b2), fmaxf(b3, b4)), fmaxf(b5, b6));
                        const bool h0 = !(f0 < 0.0f || n0 > f0 || n0 > t_min);      // NaN -> visit
                        const bool h1 = !(f1 < 0.0f || n1 > f1 || n1 > t_min);
                        if (h0 && h1) {                                            // nearer child first, the other one waits
                            const bool first0 = n0 <= n1;
                            s_stack[sp++ * 256 + tid] = first0 ? nd.ref1 : nd.ref0;   // depth-bounded by the builder
                            cur = first0 ? nd.ref0 : nd.ref1;
                            continue;
                        }
                        if (h0 || h1) { cur = h0 ? nd.ref0 : nd.ref1; continue; }
                    } else {
                        const int v = -cur - 1, first = v >> 3, cnt = v & 7;
                        for (int k = 0; k < cnt; k++) {
                            v3 tp, tn;
                            const float t = triangleTest(p.lfaces[first + k], o, d, tp, tn);
                            const int fi = p.lidx[first + k];
                            if (t > 0.0f && (t_min > t || (t_min == t && best_face >= 0 && fi < best_face))) {
                                t_min = t; materialid = p.lfaces[first + k].materialid; hitP = tp; normal = tn;
                                best_face = fi;
                            }
                        }
                    }
                    if (sp == 0) break;
                    cur = s_stack[--sp * 256 + tid];
                }
            }
        }
        const bool hit = materialid != -1;
        if (FIRST && p.cache_mode == 1) {                          // CACHE_BOUNCE, iter == 1 (:466-472)
            float* c = p.cache + i;
            c[0] = t_min; c[(size_t)P] = __int_as_float(materialid);
            c[(size_t)2 * P] = hitP.x; c[(size_t)3 * P] = hitP.y; c[(size_t)4 * P] = hitP.z;
            c[(size_t)5 * P] = normal.x; c[(size_t)6 * P] = normal.y; c[(size_t)7 * P] = normal.z;
        }
        if (p.sortkey) p.sortkey[FIRST ? i : t] = hit ? materialid : 0;   // the hit record's materialId (memset 0, :478)
        const v3 surfN = vnormalize(normal);
        const int x = i % p.W, y = i / p.W;
        const size_t gd = (size_t)y * p.stride + (size_t)(p.W - x - 1);         // h-flipped destination (:297-299)

        // ---- shadeMaterial :333-390
        int new_rem;
        if (hit) {
            uint32_t rng = make_seed(p.iter, idx, rem);
            const aipt_material m = p.mats[materialid];
            if (m.emittance > 0.0f) {
                new_rem = 0;
                col = vscale(vmul(col, V(m.color[0], m.color[1], m.color[2])), m.emittance);
            } else {
                scatterRay(o, d, col, surfN, hitP, m, rng);
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
            if (p.mat0) p.mat0[i] = materialid;
        }
        if (new_rem == 0) {
            // finalGather + copy_data (:393-402, :81-94): image += colour; planes 0-2 = image / iter.  At iter 1 the image
            // starts from zero (pathtraceInit memsets it, :102), so it is just this path's colour.
            const float fiter = (float)p.iter;
            float* gb = p.gbuf + gd;
            const size_t ii = (size_t)y * p.W + (size_t)(p.W - x - 1);
            float ax = col.x, ay = col.y, az = col.z;
            if (p.iter > 1) { ax = p.image[ii] + col.x; ay = p.image[ii + P] + col.y; az = p.image[ii + 2 * (size_t)P] + col.z; }
            p.image[ii] = ax; p.image[ii + P] = ay; p.image[ii + 2 * (size_t)P] = az;
            gb[0] = ax / fiter;
            gb[p.plane] = ay / fiter;
            gb[p.plane * 2] = az / fiter;
            remp[i] = 0;
        } else {
            ox[i] = o.x; oy[i] = o.y; oz[i] = o.z;
            dx[i] = d.x; dy[i] = d.y; dz[i] = d.z;
            cr[i] = col.x; cg[i] = col.y; cb[i] = col.z;
            remp[i] = new_rem;
            alive_after = true;
        }
    }

    // ---- live count of this workgroup for the next bounce
    const unsigned long long m2 = __ballot(alive_after);
    if (lane == 0) s_wave[wave] = __popcll(m2);
    __syncthreads();
    if (tid == 0) {
        const int c = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
        p.cnt_out[blockIdx.x] = c;
        if (c) atomicAdd(&p.n_live[p.bounce + 1], c);
    }
}

// Stable stream compaction of the live list (the wave64 ballot/popcount analogue of thrust::partition, pathtrace.cu:505):
// entry t of the input list survives iff its path still has bounces left; survivors keep their order.  Output position =
// (live counts of the lower workgroups, written by trace_bounce) + (ballot prefix inside the workgroup).
__global__ __launch_bounds__(256) void trace_compact(const TraceParams p) {
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.bounce == 0 ? p.P : p.n_live[p.bounce];
    if ((int)(blockIdx.x * 256) >= n) return;
    const int t = blockIdx.x * 256 + tid;
    const int* remp = reinterpret_cast<const int*>(p.st + (size_t)9 * p.P);
    int i = 0;
    bool alive = false;
    if (t < n) {
        i = p.live_in ? p.live_in[t] : t;
        alive = remp[i] != 0;
    }
    int part = 0;
    for (int j = tid; j < (int)blockIdx.x; j += 256) part += p.cnt_out[j];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
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
    if (alive) p.live_out[base + woff + __popcll(mask & ((1ull << lane) - 1ull))] = i;
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

void trace_destroy(aipt_ctx* ctx) {
    TraceState* s = ctx->trace;
    if (!s) return;
    hipFree(s->d_geoms); hipFree(s->d_mats); hipFree(s->d_faces); hipFree(s->d_nodes); hipFree(s->d_lfaces); hipFree(s->d_lidx);
    hipFree(s->d_state); hipFree(s->d_cnt[0]); hipFree(s->d_cnt[1]); hipFree(s->d_nlive); hipFree(s->d_mat0); hipFree(s->d_image); hipFree(s->d_live[0]); hipFree(s->d_live[1]);
    hipFree(s->d_cache); hipFree(s->d_live3); hipFree(s->d_sortkey); hipFree(s->d_hist);
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

}  // namespace aipt

using namespace aipt;

extern "C" {

int aipt_scene_upload(aipt_ctx* ctx, const aipt_geom* geoms, int ngeoms, const aipt_material* materials, int nmaterials,
                      const aipt_face* faces, int nfaces, const aipt_aabb* mesh_box) {
    AIPT_CHECK_CTX(ctx);
    if (ngeoms < 0 || nmaterials <= 0 || nfaces < 0 || (ngeoms && !geoms) || !materials || (nfaces && (!faces || !mesh_box)))
        return fail(ctx, AIPT_E_INVALID, "aipt_scene_upload: bad arguments (%d geoms, %d materials, %d faces)", ngeoms,
                    nmaterials, nfaces);
    for (int i = 0; i < ngeoms; i++)
        if (geoms[i].materialid < 0 || geoms[i].materialid >= nmaterials)
            return fail(ctx, AIPT_E_INVALID, "geom %d: material %d of %d", i, geoms[i].materialid, nmaterials);
    for (int i = 0; i < nfaces; i++)
        if (faces[i].materialid < 0 || faces[i].materialid >= nmaterials)
            return fail(ctx, AIPT_E_INVALID, "face %d: material %d of %d", i, faces[i].materialid, nmaterials);
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    TraceState* s = tstate(ctx);
    hipFree(s->d_geoms); hipFree(s->d_mats); hipFree(s->d_faces); hipFree(s->d_nodes); hipFree(s->d_lfaces); hipFree(s->d_lidx);
    s->d_geoms = nullptr; s->d_mats = nullptr; s->d_faces = nullptr; s->have_scene = false;
    s->d_nodes = nullptr; s->d_lfaces = nullptr; s->d_lidx = nullptr; s->nnodes = 0;
    std::vector<DevGeom> dg(ngeoms);
    for (int i = 0; i < ngeoms; i++) dg[i] = make_dev_geom(geoms[i]);
    s->h_geoms.assign(geoms, geoms + ngeoms);
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_geoms, sizeof(DevGeom) * (ngeoms ? ngeoms : 1)));
    if (ngeoms) AIPT_HIP(ctx, hipMemcpy(s->d_geoms, dg.data(), sizeof(DevGeom) * ngeoms, hipMemcpyHostToDevice));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_mats, sizeof(aipt_material) * nmaterials));
    AIPT_HIP(ctx, hipMemcpy(s->d_mats, materials, sizeof(aipt_material) * nmaterials, hipMemcpyHostToDevice));
    static_assert(sizeof(DevFace) == sizeof(aipt_face), "face layout");
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_faces, sizeof(DevFace) * (nfaces ? nfaces : 1)));
    if (nfaces) AIPT_HIP(ctx, hipMemcpy(s->d_faces, faces, sizeof(DevFace) * nfaces, hipMemcpyHostToDevice));
    if (nfaces) s->box = *mesh_box; else memset(&s->box, 0, sizeof(s->box));
    if (nfaces) {
        std::vector<BvhNode> nodes;
        std::vector<int> lidx;
        s->bvh_depth = build_bvh(faces, nfaces, nodes, lidx);
        if (s->bvh_depth < 0) return fail(ctx, AIPT_E_INVALID, "aipt_scene_upload: mesh of %d faces is too deep for the traversal stack", nfaces);
        std::vector<DevFace> lf(lidx.size());
        for (size_t i = 0; i < lidx.size(); i++) memcpy(&lf[i], &faces[lidx[i]], sizeof(DevFace));
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_nodes, sizeof(BvhNode) * nodes.size()));
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_lfaces, sizeof(DevFace) * lf.size()));
        AIPT_HIP(ctx, hipMalloc((void**)&s->d_lidx, sizeof(int) * lidx.size()));
        AIPT_HIP(ctx, hipMemcpy(s->d_nodes, nodes.data(), sizeof(BvhNode) * nodes.size(), hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMemcpy(s->d_lfaces, lf.data(), sizeof(DevFace) * lf.size(), hipMemcpyHostToDevice));
        AIPT_HIP(ctx, hipMemcpy(s->d_lidx, lidx.data(), sizeof(int) * lidx.size(), hipMemcpyHostToDevice));
        s->nnodes = (int)nodes.size();
    }
    s->ngeoms = ngeoms; s->nmats = nmaterials; s->nfaces = nfaces;
    s->have_scene = true; s->cache_valid = false;
    return AIPT_OK;
}

int aipt_scene_free(aipt_ctx* ctx) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    TraceState* s = tstate(ctx);
    hipFree(s->d_geoms); hipFree(s->d_mats); hipFree(s->d_faces); hipFree(s->d_nodes); hipFree(s->d_lfaces); hipFree(s->d_lidx);
    s->d_geoms = nullptr; s->d_mats = nullptr; s->d_faces = nullptr; s->have_scene = false;
    s->d_nodes = nullptr; s->d_lfaces = nullptr; s->d_lidx = nullptr; s->nnodes = 0;
    return AIPT_OK;
}

int aipt_trace_configure(aipt_ctx* ctx, int width, int height) {
    AIPT_CHECK_CTX(ctx);
    if (width <= 0 || height <= 0 || (long)width * height > (1l << 30))
        return fail(ctx, AIPT_E_INVALID, "aipt_trace_configure: %dx%d", width, height);
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    TraceState* s = tstate(ctx);
    if (s->W == width && s->H == height && s->d_state) return AIPT_OK;
    hipFree(s->d_state); hipFree(s->d_cnt[0]); hipFree(s->d_cnt[1]); hipFree(s->d_nlive); hipFree(s->d_mat0); hipFree(s->d_image);
    hipFree(s->d_live[0]); hipFree(s->d_live[1]);
    hipFree(s->d_cache); hipFree(s->d_live3); hipFree(s->d_sortkey); hipFree(s->d_hist);
    s->d_cache = nullptr; s->d_live3 = nullptr; s->d_sortkey = nullptr; s->d_hist = nullptr; s->hist_keys = 0;
    s->d_live[0] = s->d_live[1] = nullptr;
    s->d_state = nullptr; s->d_cnt[0] = s->d_cnt[1] = nullptr; s->d_nlive = nullptr; s->d_mat0 = nullptr; s->d_image = nullptr;
    const int P = width * height, nblk = (P + 255) / 256;
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_state, sizeof(float) * 10 * (size_t)P));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_cnt[0], sizeof(int) * nblk));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_cnt[1], sizeof(int) * nblk));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_nlive, sizeof(int) * (MAX_DEPTH + 1)));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_mat0, sizeof(int) * (size_t)P));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_image, sizeof(float) * 3 * (size_t)P));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_live[0], sizeof(int) * (size_t)P));
    AIPT_HIP(ctx, hipMalloc((void**)&s->d_live[1], sizeof(int) * (size_t)P));
    s->W = width; s->H = height; s->P = P; s->nblk = nblk;
    s->mat0_valid = false; s->cache_valid = false;
    return AIPT_OK;
}

int aipt_trace(aipt_ctx* ctx, const aipt_camera* cam, int iter, int depth, uint32_t flags,
               float* d_gbuf, int gbuf_rows, int gbuf_stride) {
    AIPT_CHECK_CTX(ctx);
    return aipt::trace_on_stream(ctx, ctx->stream, cam, iter, depth, flags, d_gbuf, gbuf_rows, gbuf_stride);
}

extern "C++" {
namespace aipt {
int trace_on_stream(aipt_ctx* ctx, hipStream_t st, const aipt_camera* cam, int iter, int depth, uint32_t flags,
                    float* d_gbuf, int gbuf_rows, int gbuf_stride) {
    TraceState* s = tstate(ctx);
    if (!s->have_scene) return fail(ctx, AIPT_E_STATE, "aipt_trace: no scene uploaded");
    if (!s->d_state) return fail(ctx, AIPT_E_STATE, "aipt_trace: call aipt_trace_configure first");
    if (!cam || !d_gbuf) return fail(ctx, AIPT_E_INVALID, "aipt_trace: NULL argument");
    if (cam->resolution[0] != s->W || cam->resolution[1] != s->H)
        return fail(ctx, AIPT_E_INVALID, "aipt_trace: camera is %dx%d, configured %dx%d", cam->resolution[0],
                    cam->resolution[1], s->W, s->H);
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
    if (cache && !s->d_cache) AIPT_HIP(ctx, hipMalloc((void**)&s->d_cache, sizeof(float) * 8 * (size_t)s->P));
    if (cache && iter > 1 && !s->cache_valid)
        return fail(ctx, AIPT_E_STATE, "aipt_trace: AIPT_TRACE_CACHE_FIRST_BOUNCE at iter %d without a cached iter 1", iter);
    if (sortmat) {
        if (!s->d_live3) AIPT_HIP(ctx, hipMalloc((void**)&s->d_live3, sizeof(int) * (size_t)s->P));
        if (!s->d_sortkey) AIPT_HIP(ctx, hipMalloc((void**)&s->d_sortkey, sizeof(int) * (size_t)s->P));
        if (s->hist_keys < s->nmats) {
            AIPT_HIP(ctx, hipStreamSynchronize(st));
            hipFree(s->d_hist); s->d_hist = nullptr;
            AIPT_HIP(ctx, hipMalloc((void**)&s->d_hist, sizeof(int) * (size_t)s->nmats * s->nblk));
            s->hist_keys = s->nmats;
        }
    }
    if (ctx->last_trace_stream && ctx->last_trace_stream != st) AIPT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_traced, 0));
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
    p.cam = *cam; p.iter = iter; p.trace_depth = depth; p.flags = flags;
    p.W = s->W; p.H = s->H; p.P = s->P;
    p.st = s->d_state;
    p.geoms = s->d_geoms; p.ngeoms = s->ngeoms; p.mats = s->d_mats;
    p.faces = s->d_faces; p.nfaces = s->nfaces; p.box = s->box;
    p.nodes = s->d_nodes; p.nnodes = s->nnodes; p.lfaces = s->d_lfaces; p.lidx = s->d_lidx;
    p.gbuf = d_gbuf; p.plane = (size_t)gbuf_rows * gbuf_stride; p.stride = gbuf_stride;
    p.n_live = s->d_nlive;
    p.mat0 = (flags & AIPT_TRACE_RECORD_MAT0) ? s->d_mat0 : nullptr;
    p.image = s->d_image;
    p.cache = s->d_cache; p.cache_mode = cache ? (iter == 1 ? 1 : 2) : 0;
    p.sortkey = sortmat ? s->d_sortkey : nullptr;
    p.hist = s->d_hist; p.nkeys = s->nmats; p.nblk = s->nblk;
    p.sort_in = nullptr; p.sort_out = nullptr;
    AIPT_HIP(ctx, hipMemsetAsync(s->d_nlive, 0, sizeof(int) * (MAX_DEPTH + 1), st));
    int* lists[3] = {s->d_live[0], s->d_live[1], s->d_live3};
    int cur = -1;                                               // live list the bounce reads (-1: bounce 0, all pixels)
    for (int b = 0; b < depth; b++) {
        p.bounce = b;
        p.cnt_in = nullptr;
        p.cnt_out = s->d_cnt[0];
        p.live_in = cur < 0 ? nullptr : lists[cur];
        const int nxt = cur < 0 ? 0 : (cur + 1) % (sortmat ? 3 : 2);
        p.live_out = lists[nxt];
        const bool mesh = s->nfaces > 0;
        const size_t stack_bytes = mesh ? (size_t)(s->bvh_depth + 1) * 256 * sizeof(int) : 0;
        if (b == 0 && mesh) hipLaunchKernelGGL((trace_bounce<true, true>), dim3(s->nblk), dim3(256), stack_bytes, st, p);
        else if (b == 0) hipLaunchKernelGGL((trace_bounce<true, false>), dim3(s->nblk), dim3(256), 0, st, p);
        else if (mesh) hipLaunchKernelGGL((trace_bounce<false, true>), dim3(s->nblk), dim3(256), stack_bytes, st, p);
        else hipLaunchKernelGGL((trace_bounce<false, false>), dim3(s->nblk), dim3(256), 0, st, p);
        if (b + 1 < depth) {
            hipLaunchKernelGGL(trace_compact, dim3(s->nblk), dim3(256), 0, st, p);
            cur = nxt;
            if (sortmat) {                                      // compacted list -> sorted list (thrust::sort_by_key, :508-510)
                const int srt = (cur + 1) % 3;
                p.sort_in = lists[cur]; p.sort_out = lists[srt];
                hipLaunchKernelGGL(trace_sort_hist, dim3(s->nblk), dim3(256), 0, st, p);
                hipLaunchKernelGGL(trace_sort_scan, dim3(1), dim3(1024), 0, st, p);
                hipLaunchKernelGGL(trace_sort_scatter, dim3(s->nblk), dim3(256), 0, st, p);
                cur = srt;
            }
        }
    }
    if (cache && iter == 1) s->cache_valid = true;
    AIPT_HIP(ctx, hipGetLastError());
    AIPT_HIP(ctx, hipEventRecord(ctx->ev_traced, st));
    ctx->last_trace_stream = st;
    s->last_depth = depth;
    s->mat0_valid = p.mat0 != nullptr;
    return AIPT_OK;
}
}  // namespace aipt
}  // extern "C++"

int aipt_trace_live_counts(aipt_ctx* ctx, int* h_n_live, int n) {
    AIPT_CHECK_CTX(ctx);
    TraceState* s = tstate(ctx);
    if (!s->d_nlive || !s->last_depth) return fail(ctx, AIPT_E_STATE, "aipt_trace_live_counts: no trace has run");
    if (!h_n_live || n < 1) return fail(ctx, AIPT_E_INVALID, "aipt_trace_live_counts: bad arguments");
    std::vector<int> tmp(MAX_DEPTH + 1);
    if (ctx->last_trace_stream) AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_traced, 0));
    AIPT_HIP(ctx, hipMemcpyAsync(tmp.data(), s->d_nlive, sizeof(int) * (MAX_DEPTH + 1), hipMemcpyDeviceToHost, ctx->stream));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    tmp[0] = s->P;
    for (int i = 0; i < n; i++) h_n_live[i] = i <= s->last_depth ? tmp[i] : 0;
    return AIPT_OK;
}

int aipt_trace_first_hit_materials(aipt_ctx* ctx, int* h_mat, int n) {
    AIPT_CHECK_CTX(ctx);
    TraceState* s = tstate(ctx);
    if (!s->mat0_valid) return fail(ctx, AIPT_E_STATE, "aipt_trace_first_hit_materials: last trace did not record them");
    if (!h_mat || n != s->P) return fail(ctx, AIPT_E_INVALID, "aipt_trace_first_hit_materials: n=%d, expected %d", n, s->P);
    if (ctx->last_trace_stream) AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_traced, 0));
    AIPT_HIP(ctx, hipMemcpyAsync(h_mat, s->d_mat0, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    return AIPT_OK;
}

}  // extern "C"
