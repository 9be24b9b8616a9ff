/*
 * trace_oracle.c -- CPU restatement of the reference's 1-spp path tracer (pathtrace() hot loop).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (ai_path_tracer_denoiser_amd/) may link,
 * load or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, as the checker / reported CPU baseline.
 *
 * Follows, statement by statement (paths relative to /root/reference/Inference):
 *   src/pathtrace.cu   :52-56 makeSeededRandomEngine, :81-94 copy_data, :155-182 generateRayFromCamera,
 *                      :200-306 computeIntersections, :333-390 shadeMaterial, :393-402 finalGather,
 *                      :404-411 path_termination_test + thrust::partition (:505), :422-528 pathtrace driver
 *   src/intersections.h:12-20 utilhash, :27-29 getPointOnRay, :34-36 multiplyMV, :52-94 box, :106-148 sphere,
 *                      :159-172 triangle, :175-200 RayAABBintersect
 *   src/interactions.h :13-44 calculateRandomDirectionInHemisphere, :74-85 refract, :116-120 schlick,
 *                      :170-259 scatterRay (live branch = DIELECTRIC false :6, FRESNELS true :5)
 *   src/utilities.cpp  :45-52 buildTransformationMatrix; src/scene.cpp :92-95, :143-152; src/main.cpp :66-78, :122-140
 *   external/include/glm (0.9.6.3): func_geometric.inl (dot/cross/normalize/length/reflect/refract),
 *                      gtx/intersect.inl:37-74 intersectRayTriangle, type_mat4x4.inl (mat*vec :591-626, mat*mat :685-703,
 *                      compute_inverse :37-92), gtc/matrix_transform.inl (translate :40, rotate :52, scale :122),
 *                      gtc/matrix_inverse.inl:95-147 inverseTranspose
 * Third-party, un-vendored: thrust::default_random_engine = minstd_rand (LCG a=48271, c=0, m=2^31-1) and
 *   thrust::uniform_real_distribution<float> (CUDA-toolkit Thrust, version not pinned by the reference; the identical
 *   source ships as rocThrust: /opt/rocm/include/thrust/random/detail/{linear_congruential_engine,uniform_real_distribution}.inl).
 *
 * PINNING: the reference has no tests for this path and its headers cannot be compiled here without a stand-in
 * <cuda_runtime.h> (sceneStructs.h:5), so there is no oracle/_ref build for the trace stage.  The restatement is pinned to
 * (a) the known answers SURVEY.md Appendix B / F8 recorded from the reference headers run in the survey container
 *     (utilhash/seed, minstd 10000th draw, u01 draws, calculateRandomDirectionInHemisphere, struct sizes, the F8 triangle
 *     hit-point quirk) and (b) libstdc++'s std::minstd_rand (same published LCG) -- tests/test_oracle_trace.py.
 *     Everything else is "parity unpinned" against the reference build and says so in DESIGN.md.
 *
 * Reference quirks reproduced on purpose: F6 constant seed per frame (iter is an argument), F7 AA-jitter
 * seed reads uninitialised remainingBounces -> defined here as 0, F8 mismatched barycentrics in the triangle hit point,
 * F9 depth-exhausted paths keep their colour, scatterRay's reflect-of-the-refracted-direction, camera.right not normalised.
 * Choices where the reference is unspecified: the two AA u01 draws are evaluated left to right (x jitter first);
 * unqualified min/max in device code are CUDA's fminf/fmaxf; pow(x,5) is ((x*x)*(x*x))*x; sin/cos of the hemisphere angle
 * use det_sincosf below (Cody-Waite + Cephes minimax polynomials, plain fp32 mul/add) so that the HIP kernels can match
 * this file bit for bit.  Build with -ffp-contract=off and without fast-math (oracle/Makefile).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

/* POD layouts identical to the reference's sceneStructs.h (sizes [probed] in SURVEY Appendix B). */
typedef struct {
    int type;            /* 0 = SPHERE, 1 = CUBE (sceneStructs.h:10-13) */
    int materialid;
    float translation[3], rotation[3], scale[3];
    float transform[16], inverseTransform[16], invTranspose[16];   /* column-major: m[col*4+row] */
    float vel[3];
} orc_geom;        /* 248 B */
typedef struct { float v[3][3]; float n[3][3]; int materialid; } orc_face;     /* 76 B */
typedef struct {
    float color[3]; float spec_exponent; float spec_color[3];
    float hasReflective, hasRefractive, indexOfRefraction, emittance;
} orc_material;    /* 44 B */
typedef struct {
    int res[2];
    float position[3], lookAt[3], view[3], up[3], right[3];
    float fov[2], pixelLength[2];
} orc_camera;      /* 84 B */
typedef struct { float lb[3], ub[3]; } orc_aabb;   /* 24 B */

_Static_assert(sizeof(orc_geom) == 248, "Geom");
_Static_assert(sizeof(orc_face) == 76, "Face");
_Static_assert(sizeof(orc_material) == 44, "Material");
_Static_assert(sizeof(orc_camera) == 84, "Camera");
_Static_assert(sizeof(orc_aabb) == 24, "MeshBoundingBox");

#define PI_F 3.1415926535897932384626422832795028841971f
#define TWO_PI_F 6.2831853071795864769252867665590057683943f
#define SQRT_OF_ONE_THIRD_F 0.5773502691896257645091487805019574556476f

/* ---------------------------------------------------------------- vector helpers (GLM op order) */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }   /* tmp.x+tmp.y+tmp.z */
static inline v3 vcross(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }   /* x * inversesqrt(dot) */
static inline v3 vreflect(v3 I, v3 N) { return vsub(I, vscale(vscale(N, vdot(N, I)), 2.0f)); }
static inline v3 glm_refract(v3 I, v3 N, float eta) {
    const float d = vdot(N, I);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    const v3 r = vsub(vscale(I, eta), vscale(N, eta * d + sqrtf(k)));
    return vscale(r, (float)(k >= 0.0f));
}
static inline float glm_min(float x, float y) { return x < y ? x : y; }
static inline float glm_max(float x, float y) { return x > y ? x : y; }

/* mat4 * vec4, truncated to vec3 (multiplyMV, intersections.h:34; GLM type_mat4x4.inl:618-629) */
static inline v3 mulMV(const float* m, v3 v, float w) {
    v3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + (m[8] * v.z + m[12] * w);
    r.y = (m[1] * v.x + m[5] * v.y) + (m[9] * v.z + m[13] * w);
    r.z = (m[2] * v.x + m[6] * v.y) + (m[10] * v.z + m[14] * w);
    return r;
}

/* ---------------------------------------------------------------- RNG (thrust minstd_rand + uniform_real) */
uint32_t orc_utilhash(uint32_t a) {            /* intersections.h:12-20 */
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
/* makeSeededRandomEngine (pathtrace.cu:52-56) + linear_congruential_engine::seed */
uint32_t orc_seed(int iter, int index, int depth) {
    uint32_t h = orc_utilhash((1u << 31) | ((uint32_t)depth << 22) | (uint32_t)iter) ^ orc_utilhash((uint32_t)index);
    uint32_t s = h % 2147483647u;
    return s == 0 ? 1u : s;
}
uint32_t orc_lcg_next(uint32_t* x) {
    *x = (uint32_t)(((uint64_t)(*x) * 48271ull) % 2147483647ull);
    return *x;
}
/* uniform_real_distribution<float>(a,b): (float(x - min) / (1.f + float(max - min))) * (b - a) + a, min=1, max=m-1 */
float orc_u01(uint32_t* x, float a, float b) {
    float r = (float)(orc_lcg_next(x) - 1u);
    r /= (1.0f + (float)(2147483646u - 1u));
    return (r * (b - a)) + a;
}

/* ---------------------------------------------------------------- deterministic sin/cos on [0, 2pi] */
void orc_det_sincosf(float x, float* s, float* c) {
    const int q = (int)(x * 0.636619772367581343f + 0.5f);   /* nearest multiple of pi/2 */
    const float fq = (float)q;
    float r = x - fq * 1.5703125f;                            /* Cody-Waite, pi/2 in three parts */
    r = r - fq * 4.837512969970703125e-4f;
    r = r - fq * 7.54978995489188216e-8f;
    const float z = r * r;
    /* Cephes sinf/cosf minimax polynomials on [-pi/4, pi/4] */
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
                     - 0.5f * z + 1.0f;
    switch (q & 3) {
        case 0: *s = ps;  *c = pc;  break;
        case 1: *s = pc;  *c = -ps; break;
        case 2: *s = -ps; *c = -pc; break;
        default: *s = -pc; *c = ps; break;
    }
}

/* ---------------------------------------------------------------- intersections.h */
typedef struct { v3 origin, direction; } ray_t;

static inline v3 getPointOnRay(ray_t r, float t) {       /* :27-29 */
    return vadd(r.origin, vscale(vnormalize(r.direction), t - .0001f));
}

float orc_box_test(const orc_geom* box, const float* ro, const float* rd, float* P, float* N, int* outside) {
    ray_t r = {V(ro[0], ro[1], ro[2]), V(rd[0], rd[1], rd[2])};
    ray_t q;
    q.origin = mulMV(box->inverseTransform, r.origin, 1.0f);
    q.direction = vnormalize(mulMV(box->inverseTransform, r.direction, 0.0f));
    float tmin = -1e38f, tmax = 1e38f;
    v3 tmin_n = V(0, 0, 0), tmax_n = V(0, 0, 0);
    const float qo[3] = {q.origin.x, q.origin.y, q.origin.z};
    const float qd[3] = {q.direction.x, q.direction.y, q.direction.z};
    for (int xyz = 0; xyz < 3; ++xyz) {
        const float qdxyz = qd[xyz];
        const float t1 = (-0.5f - qo[xyz]) / qdxyz;
        const float t2 = (+0.5f - qo[xyz]) / qdxyz;
        const float ta = glm_min(t1, t2);
        const float tb = glm_max(t1, t2);
        float n[3] = {0, 0, 0};
        n[xyz] = t2 < t1 ? +1.0f : -1.0f;
        if (ta > 0 && ta > tmin) { tmin = ta; tmin_n = V(n[0], n[1], n[2]); }
        if (tb < tmax) { tmax = tb; tmax_n = V(n[0], n[1], n[2]); }
    }
    if (tmax >= tmin && tmax > 0) {
        *outside = 1;
        if (tmin <= 0) { tmin = tmax; tmin_n = tmax_n; *outside = 0; }
        const v3 ip = mulMV(box->transform, getPointOnRay(q, tmin), 1.0f);
        const v3 nn = vnormalize(mulMV(box->transform, tmin_n, 0.0f));
        P[0] = ip.x; P[1] = ip.y; P[2] = ip.z;
        N[0] = nn.x; N[1] = nn.y; N[2] = nn.z;
        return vlength(vsub(r.origin, ip));
    }
    return -1;
}

float orc_sphere_test(const orc_geom* sp, const float* ro_, const float* rd_, float* P, float* N, int* outside) {
    ray_t r = {V(ro_[0], ro_[1], ro_[2]), V(rd_[0], rd_[1], rd_[2])};
    const float radius = .5f;
    ray_t rt;
    rt.origin = mulMV(sp->inverseTransform, r.origin, 1.0f);
    rt.direction = vnormalize(mulMV(sp->inverseTransform, r.direction, 0.0f));
    const float vDotDirection = vdot(rt.origin, rt.direction);
    const float radicand = vDotDirection * vDotDirection - (vdot(rt.origin, rt.origin) - radius * radius);
    if (radicand < 0) return -1;
    const float squareRoot = sqrtf(radicand);
    const float firstTerm = -vDotDirection;
    const float t1 = firstTerm + squareRoot;
    const float t2 = firstTerm - squareRoot;
    float t;
    if (t1 < 0 && t2 < 0) return -1;
    else if (t1 > 0 && t2 > 0) { t = fminf(t1, t2); *outside = 1; }
    else { t = fmaxf(t1, t2); *outside = 0; }
    const v3 obj = getPointOnRay(rt, t);
    const v3 ip = mulMV(sp->transform, obj, 1.0f);
    v3 nn = vnormalize(mulMV(sp->invTranspose, obj, 0.0f));
    if (!*outside) nn = vneg(nn);
    P[0] = ip.x; P[1] = ip.y; P[2] = ip.z;
    N[0] = nn.x; N[1] = nn.y; N[2] = nn.z;
    return vlength(vsub(r.origin, ip));
}

/* glm::intersectRayTriangle (gtx/intersect.inl:37-74).  baryPosition is written component by component, so an early
 * return leaves the later components as the caller passed them (the reference passes an uninitialised vec3,
 * intersections.h:163). */
static int glm_intersect_ray_triangle(v3 orig, v3 dir, v3 v0, v3 v1, v3 v2, v3* bary) {
    const v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    const v3 p = vcross(dir, e2);
    const float a = vdot(e1, p);
    if (a < FLT_EPSILON) return 0;
    const float ff = 1.0f / a;
    const v3 s = vsub(orig, v0);
    bary->x = ff * vdot(s, p);
    if (bary->x < 0.0f) return 0;
    if (bary->x > 1.0f) return 0;
    const v3 q = vcross(s, e1);
    bary->y = ff * vdot(dir, q);
    if (bary->y < 0.0f) return 0;
    if (bary->y + bary->x > 1.0f) return 0;
    bary->z = ff * vdot(e2, q);
    return bary->z >= 0.0f;
}

/* triangleIntersectionTest (intersections.h:159-172) */
float orc_triangle_test(const orc_face* f, const float* ro, const float* rd, float* P, float* N) {
    const v3 orig = V(ro[0], ro[1], ro[2]), dir = V(rd[0], rd[1], rd[2]);
    const v3 v0 = V(f->v[0][0], f->v[0][1], f->v[0][2]);
    const v3 v1 = V(f->v[1][0], f->v[1][1], f->v[1][2]);
    const v3 v2 = V(f->v[2][0], f->v[2][1], f->v[2][2]);
    v3 bary = V(0, 0, 0);
    if (!glm_intersect_ray_triangle(orig, dir, v0, v1, v2, &bary)) return -1;
    const float bx = bary.x, by = bary.y, bz = bary.z;
    const float bw = 1.0f - bx - by;
    /* F8: the hit point weights v0,v1,v2 with (x, y, 1-x-y) although GLM's (x,y) weight v1,v2 */
    const v3 ip = vadd(vadd(vscale(v0, bx), vscale(v1, by)), vscale(v2, bw));
    const v3 n0 = V(f->n[0][0], f->n[0][1], f->n[0][2]);
    const v3 n1 = V(f->n[1][0], f->n[1][1], f->n[1][2]);
    const v3 n2 = V(f->n[2][0], f->n[2][1], f->n[2][2]);
    const v3 nn = vnormalize(vadd(vadd(vscale(n0, bw), vscale(n1, bx)), vscale(n2, by)));
    P[0] = ip.x; P[1] = ip.y; P[2] = ip.z;
    N[0] = nn.x; N[1] = nn.y; N[2] = nn.z;
    return bz;
}

int orc_ray_aabb(const float* ro, const float* rd, const orc_aabb* bb) {    /* intersections.h:175-200 */
    const float dx = 1.0f / rd[0], dy = 1.0f / rd[1], dz = 1.0f / rd[2];
    const float t1 = (bb->lb[0] - ro[0]) * dx, t2 = (bb->ub[0] - ro[0]) * dx;
    const float t3 = (bb->lb[1] - ro[1]) * dy, t4 = (bb->ub[1] - ro[1]) * dy;
    const float t5 = (bb->lb[2] - ro[2]) * dz, t6 = (bb->ub[2] - ro[2]) * dz;
    const float tmin = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
    const float tmax = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
    if (tmax < 0) return 0;
    if (tmin > tmax) return 0;
    return 1;
}

/* ---------------------------------------------------------------- interactions.h */
void orc_hemisphere(const float* n_, uint32_t* rng, float* out) {    /* :13-44 */
    const v3 normal = V(n_[0], n_[1], n_[2]);
    const float up = sqrtf(orc_u01(rng, 0.0f, 1.0f));
    const float over = sqrtf(1 - up * up);
    const float around = orc_u01(rng, 0.0f, 1.0f) * TWO_PI_F;
    v3 dnn;
    if (fabsf(normal.x) < SQRT_OF_ONE_THIRD_F) dnn = V(1, 0, 0);
    else if (fabsf(normal.y) < SQRT_OF_ONE_THIRD_F) dnn = V(0, 1, 0);
    else dnn = V(0, 0, 1);
    const v3 p1 = vnormalize(vcross(normal, dnn));
    const v3 p2 = vnormalize(vcross(normal, p1));
    float sn, cs;
    orc_det_sincosf(around, &sn, &cs);
    const v3 r = vadd(vadd(vscale(normal, up), vscale(p1, cs * over)), vscale(p2, sn * over));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

static int ref_refract(v3 v, v3 n, float ni_over_nt, v3* refracted) {      /* :74-85 */
    const v3 uv = vnormalize(v);
    const float dt = vdot(uv, n);
    const float discriminat = (float)(1.0 - (double)(ni_over_nt * ni_over_nt * (1 - dt * dt)));
    if (discriminat > 0) {
        *refracted = vsub(vscale(vsub(uv, vscale(n, dt)), ni_over_nt), vscale(n, sqrtf(discriminat)));
        return 1;
    }
    return 0;
}
static float schlick(float cosine, float ref_idx) {                        /* :116-120 */
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    const float x = 1 - cosine;
    const float x2 = x * x;
    return r0 + (1 - r0) * ((x2 * x2) * x);
}

typedef struct {
    v3 origin, direction, color;
    int pixelIndex, remainingBounces;
} path_t;
typedef struct {
    float t; v3 surfaceNormal; int materialId; v3 intersect;
} hit_t;

/* ---- the compile-time branches of interactions.h that the reference ships switched off, as run-time flags of
 * orc_pathtrace_ex: DIELECTRIC (:6, :179-192) = flag 1024, MESH_NORMAL_VIEW (:4) = flag 2048 */
static int g_dielectric = 0, g_normal_view = 0;
#define REF_EPSILON 0.0001f                                                 /* utilities.h:16 */

static float FresnelDielectric_Evaluate(float cosThetaI, float etaI, float etaT) {   /* :88-115 */
    cosThetaI = glm_min(glm_max(cosThetaI, -1.0f), 1.0f);                   /* glm::clamp = min(max(x, lo), hi) */
    const int entering = cosThetaI > 0.0f;
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
static void SpecularReflection_BxDF(path_t* ps, v3 intersect, v3 normal, const orc_material* m) {   /* :121-125 */
    ps->color = vmul(ps->color, V(m->spec_color[0], m->spec_color[1], m->spec_color[2]));
    ps->direction = vreflect(ps->direction, normal);
    ps->origin = vadd(intersect, vscale(ps->direction, .001f));
}
static void SpecularRefraction_BxDF(path_t* ps, v3 intersect, v3 normal, const orc_material* m) {   /* :127-146 */
    const v3 wo = ps->direction;
    const int leaving = vdot(wo, normal) > 0.f;
    const v3 n = vscale(normal, leaving ? -1.f : 1.f);
    const float eta = leaving ? m->indexOfRefraction : (1.f / m->indexOfRefraction);
    v3 wi = glm_refract(wo, n, eta);
    if (vlength(wi) < .01f) {                                               /* total internal reflection */
        ps->color = vscale(ps->color, 0.0f);
        wi = vreflect(wo, normal);
    }
    ps->color = vmul(ps->color, V(m->spec_color[0], m->spec_color[1], m->spec_color[2]));
    ps->direction = wi;
    ps->origin = vadd(intersect, vscale(ps->direction, .001f));
}
static void Glass_BxDF(path_t* ps, v3 intersect, v3 normal, const orc_material* m, uint32_t* rng) {   /* :148-163 */
    const float VdotN = vdot(vneg(ps->direction), normal);
    const int leaving = VdotN < 0.f;
    const float eI = leaving ? m->indexOfRefraction : 1.f;
    const float eT = leaving ? 1.f : m->indexOfRefraction;
    const float fresnel = FresnelDielectric_Evaluate(VdotN, eI, eT) / fabsf(VdotN);
    if (orc_u01(rng, 0.0f, 1.0f) < fresnel) SpecularReflection_BxDF(ps, intersect, normal, m);
    else SpecularRefraction_BxDF(ps, intersect, normal, m);
}
static void Lambert_BxDF(path_t* ps, v3 intersect, v3 normal, const orc_material* m, uint32_t* rng) {   /* :164-168 */
    const v3 nn = vnormalize(normal);
    const float nrm[3] = {nn.x, nn.y, nn.z};
    float h[3];
    orc_hemisphere(nrm, rng, h);
    ps->direction = V(h[0], h[1], h[2]);
    ps->color = vmul(ps->color, V(m->color[0], m->color[1], m->color[2]));
    ps->origin = vadd(intersect, vscale(ps->direction, .001f));
}

static void scatterRay(path_t* ps, const hit_t* isect, const orc_material* m, uint32_t* rng) {   /* :170-259 */
    v3 dir = ps->direction;
    v3 color = V(1.0f, 1.0f, 1.0f);
    const v3 mcolor = g_normal_view ? isect->surfaceNormal : V(m->color[0], m->color[1], m->color[2]);           /* :222-255 */
    const v3 scolor = g_normal_view ? isect->surfaceNormal : V(m->spec_color[0], m->spec_color[1], m->spec_color[2]);
    float reflective_prob = m->hasReflective;
    if (g_dielectric) {                                                     /* :179-192 */
        if (m->hasReflective > REF_EPSILON && m->hasRefractive > REF_EPSILON) Glass_BxDF(ps, isect->intersect, isect->surfaceNormal, m, rng);
        else if (m->hasReflective > REF_EPSILON) SpecularReflection_BxDF(ps, isect->intersect, isect->surfaceNormal, m);
        else if (m->hasRefractive > REF_EPSILON) SpecularRefraction_BxDF(ps, isect->intersect, isect->surfaceNormal, m);
        else Lambert_BxDF(ps, isect->intersect, isect->surfaceNormal, m, rng);
        return;
    }
    if (reflective_prob != 0 || m->hasRefractive != 0) {
        const float pdf = orc_u01(rng, 0.0f, 1.0f);
        float refrac_index_ratio, cosine;
        v3 normal;
        cosine = vdot(vnormalize(dir), isect->surfaceNormal);
        if (cosine <= 0) {
            normal = isect->surfaceNormal;
            refrac_index_ratio = 1 / m->indexOfRefraction;
            cosine = -cosine;
        } else {
            normal = vneg(isect->surfaceNormal);
            refrac_index_ratio = m->indexOfRefraction;
        }
        if (ref_refract(ps->direction, normal, refrac_index_ratio, &dir))   /* NB: overwrites dir (reference quirk) */
            reflective_prob = schlick(cosine, refrac_index_ratio);
        else
            reflective_prob = 1.0f;
        if (pdf < reflective_prob) {
            dir = vnormalize(vreflect(dir, isect->surfaceNormal));
            color = scolor;
        } else {
            dir = vnormalize(glm_refract(ps->direction, normal, refrac_index_ratio));
            if (!vlength(dir)) {
                dir = vnormalize(vreflect(dir, isect->surfaceNormal));
                color = scolor;
            } else
                color = mcolor;
        }
    } else {
        float h[3];
        const float nrm[3] = {isect->surfaceNormal.x, isect->surfaceNormal.y, isect->surfaceNormal.z};
        orc_hemisphere(nrm, rng, h);
        dir = vnormalize(V(h[0], h[1], h[2]));
        color = mcolor;
    }
    ps->direction = dir;
    ps->origin = vadd(isect->intersect, vscale(dir, 0.01f));
    if (g_normal_view) color = V(fabsf(color.x), fabsf(color.y), fabsf(color.z));   /* :254 */
    ps->color = vmul(ps->color, color);
}

/* test hooks of the DIELECTRIC branch: the Fresnel reflectance; one scatterRay call with the branch switched on */
float orc_fresnel_dielectric(float cosThetaI, float etaI, float etaT) { return FresnelDielectric_Evaluate(cosThetaI, etaI, etaT); }
void orc_scatter(float* io, const float* hit, const orc_material* m, uint32_t* rng);
void orc_scatter_dielectric(float* io, const float* hit, const orc_material* m, uint32_t* rng) {
    const int keep = g_dielectric;
    g_dielectric = 1;
    orc_scatter(io, hit, m, rng);
    g_dielectric = keep;
}

/* test hook: one scatterRay call on flat arrays; io = origin[3] dir[3] color[3]; hit = t, n[3], P[3] */
void orc_scatter(float* io, const float* hit, const orc_material* m, uint32_t* rng) {
    path_t p = {V(io[0], io[1], io[2]), V(io[3], io[4], io[5]), V(io[6], io[7], io[8]), 0, 1};
    hit_t h = {hit[0], V(hit[1], hit[2], hit[3]), 0, V(hit[4], hit[5], hit[6])};
    scatterRay(&p, &h, m, rng);
    io[0] = p.origin.x; io[1] = p.origin.y; io[2] = p.origin.z;
    io[3] = p.direction.x; io[4] = p.direction.y; io[5] = p.direction.z;
    io[6] = p.color.x; io[7] = p.color.y; io[8] = p.color.z;
}

/* ---------------------------------------------------------------- pathtrace.cu kernels */
static void generateRay(const orc_camera* cam, int iter, int traceDepth, int x, int y, int aa, path_t* seg) {   /* :155-182 */
    const int index = x + (y * cam->res[0]);
    uint32_t rng = orc_seed(iter, index, 0 /* F7: uninitialised in the reference; defined as 0 */);
    const v3 view = V(cam->view[0], cam->view[1], cam->view[2]);
    const v3 right = V(cam->right[0], cam->right[1], cam->right[2]);
    const v3 up = V(cam->up[0], cam->up[1], cam->up[2]);
    seg->origin = V(cam->position[0], cam->position[1], cam->position[2]);
    seg->color = V(1.0f, 1.0f, 1.0f);
    const float jx = orc_u01(&rng, -0.5f, 0.5f);     /* AA true (pathtrace.cu:25) */
    const float jy = orc_u01(&rng, -0.5f, 0.5f);
    float sx = (float)x - (float)cam->res[0] * 0.5f;
    float sy = (float)y - (float)cam->res[1] * 0.5f;
    if (aa) { sx = sx + jx; sy = sy + jy; }                /* :170-180 */
    seg->direction = vnormalize(vsub(vsub(view, vscale(vscale(right, cam->pixelLength[0]), sx)),
                                     vscale(vscale(up, cam->pixelLength[1]), sy)));
    seg->pixelIndex = index;
    seg->remainingBounces = traceDepth;
}

/* trace_bvh.c: the same nearest face as the loop below, through a CPU BVH (flag 256 of orc_pathtrace_ex) */
void orc_bvh_prepare(const orc_face* faces, int nfaces);
void orc_bvh_nearest(const orc_face* faces, int nfaces, const float* ro, const float* rd, float* t_min, int* best,
                     float* P, float* N);
static int g_use_bvh = 0, g_no_cull = 0;

static void computeIntersection(const path_t* ps, const orc_geom* geoms, int ngeoms, const orc_face* faces, int nfaces,
                                const orc_aabb* box, hit_t* out, v3* raw_normal) {   /* :200-306 */
    float t_min = FLT_MAX;
    int materialid = -1;
    v3 ip = V(0, 0, 0), normal = V(0, 0, 0);
    const float ro[3] = {ps->origin.x, ps->origin.y, ps->origin.z};
    const float rd[3] = {ps->direction.x, ps->direction.y, ps->direction.z};
    float P[3], N[3];
    int outside = 1;
    for (int i = 0; i < ngeoms; i++) {
        float t = -1;
        if (geoms[i].type == 1) t = orc_box_test(&geoms[i], ro, rd, P, N, &outside);
        else if (geoms[i].type == 0) t = orc_sphere_test(&geoms[i], ro, rd, P, N, &outside);
        if (t > 0.0f && t_min > t) {
            t_min = t; materialid = geoms[i].materialid;
            ip = V(P[0], P[1], P[2]); normal = V(N[0], N[1], N[2]);
        }
    }
    int walked = 0;
    if (g_use_bvh && nfaces && (g_no_cull || orc_ray_aabb(ro, rd, box))) {
        int best = -1;
        float tm = t_min;
        orc_bvh_nearest(faces, nfaces, ro, rd, &tm, &best, P, N);
        if (best >= 0) { t_min = tm; materialid = faces[best].materialid; ip = V(P[0], P[1], P[2]); normal = V(N[0], N[1], N[2]); }
        walked = best != -2;                        /* -2: tree not prepared for this array -> the exhaustive loop */
    }
    if (!walked && nfaces && (g_no_cull || orc_ray_aabb(ro, rd, box))) {      /* RAY_CULLING true (:23, :258); false (:270-281) = flag 512 */
        for (int i = 0; i < nfaces; i++) {
            const float t = orc_triangle_test(&faces[i], ro, rd, P, N);
            if (t > 0.0f && t_min > t) {
                t_min = t; materialid = faces[i].materialid;
                ip = V(P[0], P[1], P[2]); normal = V(N[0], N[1], N[2]);
            }
        }
    }
    if (materialid == -1) {
        out->t = -1.0f; out->materialId = 0; out->surfaceNormal = V(0, 0, 0); out->intersect = V(0, 0, 0);
    } else {
        out->t = t_min; out->materialId = materialid;
        out->surfaceNormal = vnormalize(normal);
        out->intersect = ip;
    }
    *raw_normal = normal;
}

/* test hook: nearest hit of one ray; returns t (or -1), fills mat/normal/point */
float orc_intersect_scene(const float* ro, const float* rd, const orc_geom* geoms, int ngeoms, const orc_face* faces,
                          int nfaces, const orc_aabb* box, int* mat, float* n_out, float* p_out) {
    path_t ps = {V(ro[0], ro[1], ro[2]), V(rd[0], rd[1], rd[2]), V(1, 1, 1), 0, 1};
    hit_t h; v3 rn;
    computeIntersection(&ps, geoms, ngeoms, faces, nfaces, box, &h, &rn);
    *mat = h.materialId;
    n_out[0] = h.surfaceNormal.x; n_out[1] = h.surfaceNormal.y; n_out[2] = h.surfaceNormal.z;
    p_out[0] = h.intersect.x; p_out[1] = h.intersect.y; p_out[2] = h.intersect.z;
    return h.t;
}

/*
 * One frame of pathtrace() (pathtrace.cu:422-528) at 1 spp.
 *   gbuf       : float[10][Hp][W] with Hp >= H rows; planes are written for the first H rows only, the caller
 *                zero-fills (the reference cudaMemsets dev_tensor in pathtraceInit, :118-119; F6 re-inits per frame).
 *   n_live     : int[depth+1], number of live paths entering each bounce (n_live[0] = W*H) then the final count.
 *   mat0       : optional int[W*H] first-hit material id per pixel index (-1 = miss) -- integer parity channel.
 * Returns the number of bounces executed.
 */
int orc_pathtrace_accum(const orc_camera* cam, const orc_geom* geoms, int ngeoms, const orc_material* mats, int nmats,
                        const orc_face* faces, int nfaces, const orc_aabb* box, int iter, int traceDepth,
                        float* gbuf, int Hp, int* n_live, int* mat0, float* accum);

int orc_pathtrace(const orc_camera* cam, const orc_geom* geoms, int ngeoms, const orc_material* mats, int nmats,
                  const orc_face* faces, int nfaces, const orc_aabb* box, int iter, int traceDepth,
                  float* gbuf, int Hp, int* n_live, int* mat0) {
    return orc_pathtrace_accum(cam, geoms, ngeoms, mats, nmats, faces, nfaces, box, iter, traceDepth, gbuf, Hp, n_live,
                               mat0, NULL);
}

/* accum: optional float[3*W*H] = the reference's dev_image (glm::vec3 per pixel, pathtrace.cu:101-102), kept by the caller
 * across iterations 1..n of a multi-sample render (image += colour each iteration, planes 0-2 = image / iter,
 * pathtrace.cu:400, 88-92); planes 3-9 are only written at iter == 1 (:295, :379). */
int orc_pathtrace_ex(const orc_camera* cam, const orc_geom* geoms, int ngeoms, const orc_material* mats, int nmats,
                     const orc_face* faces, int nfaces, const orc_aabb* box, int iter, int traceDepth, unsigned flags,
                     float* gbuf, int Hp, int* n_live, int* mat0, float* accum, void* cache);

int orc_pathtrace_accum(const orc_camera* cam, const orc_geom* geoms, int ngeoms, const orc_material* mats, int nmats,
                        const orc_face* faces, int nfaces, const orc_aabb* box, int iter, int traceDepth,
                        float* gbuf, int Hp, int* n_live, int* mat0, float* accum) {
    return orc_pathtrace_ex(cam, geoms, ngeoms, mats, nmats, faces, nfaces, box, iter, traceDepth, 1u | 2u, gbuf, Hp,
                            n_live, mat0, accum, NULL);
}

/* The reference's compile-time switches (pathtrace.cu:20-26) as run-time flags, same bit values as include/aiptd.h:
 *   1  AA                 jitter the primary rays (:167-176)
 *   2  STREAM_COMPACTION  thrust::partition after every bounce (:504-507)
 *  32  SORT_MATERIAL      thrust::sort_by_key(dev_intersections, dev_intersections + num_paths, dev_paths, sort_cmp) (:508-510).
 *                         As written it runs AFTER the partition with the post-partition num_paths, so the key of array slot
 *                         j is the material id of the hit record the PRE-partition slot j holds (a miss has id 0: the records
 *                         are memset before every computeIntersections, :478, and a miss only sets t, :283-285).  Thrust's
 *                         sort_by_key with a user comparator is a stable merge sort (thrust/system/cuda/detail/sort.h:
 *                         sort_by_key -> stable_sort_by_key; rocThrust likewise), so the result is deterministic: a stable
 *                         sort of the surviving paths by those keys.  The next bounce seeds each path's RNG with its
 *                         slot after the sort (:351).
 * 256  (oracle only)      walk a CPU BVH instead of looping over all faces: same result (trace_bvh.c), so that full-size
 *                         frames can be checked and timed
 *  64  CACHE_BOUNCE       iter == 1: the bounce-0 hit records are saved (:466-472); iter > 1: bounce 0 reuses them instead of
 *                         intersecting (:473-476).  Only legal with AA off (assert :435).  cache = caller-held P * 36 bytes.
 */
typedef struct { int key, pos; } sort_item;
static int sort_item_cmp(const void* a, const void* b) {
    const sort_item* x = (const sort_item*)a; const sort_item* y = (const sort_item*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

int orc_pathtrace_ex(const orc_camera* cam, const orc_geom* geoms, int ngeoms, const orc_material* mats, int nmats,
                     const orc_face* faces, int nfaces, const orc_aabb* box, int iter, int traceDepth, unsigned flags,
                     float* gbuf, int Hp, int* n_live, int* mat0, float* accum, void* cache) {
    const int W = cam->res[0], H = cam->res[1];
    const int P = W * H;
    const size_t plane = (size_t)W * Hp;
    const int aa = (flags & 1u) != 0, compact = (flags & 2u) != 0, sortmat = (flags & 32u) != 0;
    const int use_cache = (flags & 64u) != 0 && cache != NULL;
    g_use_bvh = (flags & 256u) != 0;                /* test-side acceleration of the face loop, same result (trace_bvh.c) */
    g_no_cull = (flags & 512u) != 0;                /* RAY_CULLING false */
    g_dielectric = (flags & 1024u) != 0;            /* DIELECTRIC true */
    g_normal_view = (flags & 2048u) != 0;           /* MESH_NORMAL_VIEW true */
    if (g_use_bvh) orc_bvh_prepare(faces, nfaces);
    path_t* paths = (path_t*)malloc(sizeof(path_t) * P);
    path_t* tmp = (path_t*)malloc(sizeof(path_t) * P);
    hit_t* hits = (hit_t*)malloc(sizeof(hit_t) * P);
    sort_item* items = sortmat ? (sort_item*)malloc(sizeof(sort_item) * P) : NULL;
    v3* image = accum ? (v3*)accum : (v3*)calloc(P, sizeof(v3));
    (void)nmats;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) generateRay(cam, iter, traceDepth, i % W, i / W, aa, &paths[i]);

    int depth = 0, num_paths = P, done = 0;
    while (!done) {
        n_live[depth] = num_paths;
        if (depth == 0 && use_cache && iter > 1) {
            memcpy(hits, cache, sizeof(hit_t) * P);                  /* :473-476 */
        } else {
#pragma omp parallel for schedule(dynamic, 256)
            for (int idx = 0; idx < num_paths; idx++) {
                v3 rawn;
                computeIntersection(&paths[idx], geoms, ngeoms, faces, nfaces, box, &hits[idx], &rawn);
                if (depth == 0 && iter == 1) {
                    if (mat0) mat0[paths[idx].pixelIndex] = hits[idx].t >= 0 ? hits[idx].materialId : -1;
                    if (hits[idx].t >= 0) {                              /* :295-304 */
                        const int col = idx % W, row = idx / W;
                        const size_t d = (size_t)(W - col - 1) + (size_t)row * W;
                        gbuf[plane * 3 + d] = rawn.x;
                        gbuf[plane * 4 + d] = rawn.y;
                        gbuf[plane * 5 + d] = rawn.z;
                        gbuf[plane * 6 + d] = hits[idx].t;
                    }
                }
            }
            if (depth == 0 && use_cache && iter == 1) memcpy(cache, hits, sizeof(hit_t) * P);   /* :466-472 */
        }
#pragma omp parallel for schedule(static)
        for (int idx = 0; idx < num_paths; idx++) {                  /* shadeMaterial :333-390 */
            path_t* ps = &paths[idx];
            if (ps->remainingBounces == 0) continue;
            const hit_t isect = hits[idx];
            if (isect.t > 0.0f) {
                uint32_t rng = orc_seed(iter, idx, ps->remainingBounces);
                const orc_material* m = &mats[isect.materialId];
                if (m->emittance > 0.0f) {
                    ps->remainingBounces = 0;
                    ps->color = vscale(vmul(ps->color, V(m->color[0], m->color[1], m->color[2])), m->emittance);
                } else {
                    scatterRay(ps, &isect, m, &rng);
                    --ps->remainingBounces;
                }
            } else {
                ps->color = V(0, 0, 0);
                ps->remainingBounces = 0;
            }
            if (depth == 0 && iter == 1 && isect.t >= 0) {           /* :379-387 */
                const int col = idx % W, row = idx / W;
                const size_t d = (size_t)(W - col - 1) + (size_t)row * W;
                gbuf[plane * 7 + d] = ps->color.x;
                gbuf[plane * 8 + d] = ps->color.y;
                gbuf[plane * 9 + d] = ps->color.z;
            }
        }
        depth++;
        /* thrust::partition (:505): survivors keep their relative order (rocThrust partition.h:617,655-657 --
         * rejected items are emitted reversed; their order is irrelevant, finalGather scatters by pixelIndex). */
        if (compact) {
            int a = 0, b = num_paths;
            for (int i = 0; i < num_paths; i++) {
                if (paths[i].remainingBounces > 0) tmp[a++] = paths[i];
                else tmp[--b] = paths[i];
            }
            memcpy(paths, tmp, sizeof(path_t) * num_paths);
            num_paths = a;
        }
        if (sortmat && num_paths > 1) {                              /* :508-510, see the flag table above */
            for (int i = 0; i < num_paths; i++) { items[i].key = hits[i].materialId; items[i].pos = i; }
            qsort(items, num_paths, sizeof(sort_item), sort_item_cmp);
            for (int i = 0; i < num_paths; i++) tmp[i] = paths[items[i].pos];
            memcpy(paths, tmp, sizeof(path_t) * num_paths);
        }
        done = (num_paths == 0 || depth == traceDepth);
    }
    n_live[depth] = num_paths;
    for (int i = 0; i < P; i++) {                                    /* finalGather :393-402 (F9) */
        v3* px = &image[paths[i].pixelIndex];
        *px = vadd(*px, paths[i].color);
    }
    const float fiter = (float)iter;
    for (int y = 0; y < H; y++)                                      /* copy_data :81-94 */
        for (int x = 0; x < W; x++) {
            const v3 pix = image[(W - x - 1) + y * W];
            const size_t d = (size_t)x + (size_t)y * W;
            gbuf[d] = pix.x / fiter;
            gbuf[plane + d] = pix.y / fiter;
            gbuf[plane * 2 + d] = pix.z / fiter;
        }
    free(paths); free(tmp); free(hits); free(items);
    if (!accum) free(image);
    return depth;
}

/* moveGeom (pathtrace.cu:318-331), called every 4th iteration with dt = 0.10 when MOTION_BLUR (:442-446): primitives with a
 * non-zero velocity are translated and their three matrices rebuilt (device-side buildTransformationMatrix :307-316 = the
 * host one, utilities.cpp:45-52). */
void orc_build_geom(orc_geom* g);
void orc_move_geoms(orc_geom* geoms, int ngeoms, float dt) {
    for (int i = 0; i < ngeoms; i++) {
        orc_geom* g = &geoms[i];
        if (g->vel[0] == 0.0f && g->vel[1] == 0.0f && g->vel[2] == 0.0f) continue;
        for (int a = 0; a < 3; a++) g->translation[a] += g->vel[a] * dt;
        orc_build_geom(g);
    }
}

/* ---------------------------------------------------------------- host-side scene math */
static void mat_identity(float* m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }
static void mat_mul(const float* a, const float* b, float* r) {          /* type_mat4x4.inl:685-703 */
    float t[16];
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++)
            t[c * 4 + row] = ((a[0 + row] * b[c * 4 + 0] + a[4 + row] * b[c * 4 + 1]) + a[8 + row] * b[c * 4 + 2])
                             + a[12 + row] * b[c * 4 + 3];
    memcpy(r, t, 64);
}
static void mat_rotate(const float* m, float angle, v3 v, float* r) {    /* gtc/matrix_transform.inl:52-85 */
    const float c = cosf(angle), s = sinf(angle);
    const v3 axis = vnormalize(v);
    const v3 temp = vscale(axis, 1.0f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = 0 + temp.x * axis.y + s * axis.z;
    R[0][2] = 0 + temp.x * axis.z - s * axis.y;
    R[1][0] = 0 + temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = 0 + temp.y * axis.z + s * axis.x;
    R[2][0] = 0 + temp.z * axis.x + s * axis.y;
    R[2][1] = 0 + temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    float t[16];
    for (int j = 0; j < 3; j++)
        for (int row = 0; row < 4; row++)
            t[j * 4 + row] = (m[0 + row] * R[j][0] + m[4 + row] * R[j][1]) + m[8 + row] * R[j][2];
    for (int row = 0; row < 4; row++) t[12 + row] = m[12 + row];
    memcpy(r, t, 64);
}
static void mat_inverse(const float* m, float* out) {                    /* type_mat4x4.inl:37-92 */
#define M(c, r) m[(c) * 4 + (r)]
    const float Coef00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    const float Coef02 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    const float Coef03 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    const float Coef04 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    const float Coef06 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    const float Coef07 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    const float Coef08 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    const float Coef10 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    const float Coef11 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    const float Coef12 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    const float Coef14 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    const float Coef15 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    const float Coef16 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    const float Coef18 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    const float Coef19 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    const float Coef20 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    const float Coef22 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    const float Coef23 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    const float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
    const float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    const float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
    const float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    const float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
    const float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    const float Vec0[4] = {M(1,0), M(0,0), M(0,0), M(0,0)};
    const float Vec1[4] = {M(1,1), M(0,1), M(0,1), M(0,1)};
    const float Vec2[4] = {M(1,2), M(0,2), M(0,2), M(0,2)};
    const float Vec3[4] = {M(1,3), M(0,3), M(0,3), M(0,3)};
    static const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
    float Inv[16];
    for (int i = 0; i < 4; i++) {
        Inv[0 * 4 + i] = ((Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i]) * SignA[i];
        Inv[1 * 4 + i] = ((Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i]) * SignB[i];
        Inv[2 * 4 + i] = ((Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i]) * SignA[i];
        Inv[3 * 4 + i] = ((Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i]) * SignB[i];
    }
    const float d0 = M(0,0) * Inv[0], d1 = M(0,1) * Inv[4], d2 = M(0,2) * Inv[8], d3 = M(0,3) * Inv[12];
    const float Dot1 = (d0 + d1) + (d2 + d3);
    const float OneOverDeterminant = 1.0f / Dot1;
    for (int i = 0; i < 16; i++) out[i] = Inv[i] * OneOverDeterminant;
#undef M
}
static void mat_inverse_transpose(const float* m, float* out) {          /* gtc/matrix_inverse.inl:95-147 */
#define M(c, r) m[(c) * 4 + (r)]
    const float S00 = M(2,2) * M(3,3) - M(3,2) * M(2,3);
    const float S01 = M(2,1) * M(3,3) - M(3,1) * M(2,3);
    const float S02 = M(2,1) * M(3,2) - M(3,1) * M(2,2);
    const float S03 = M(2,0) * M(3,3) - M(3,0) * M(2,3);
    const float S04 = M(2,0) * M(3,2) - M(3,0) * M(2,2);
    const float S05 = M(2,0) * M(3,1) - M(3,0) * M(2,1);
    const float S06 = M(1,2) * M(3,3) - M(3,2) * M(1,3);
    const float S07 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    const float S08 = M(1,1) * M(3,2) - M(3,1) * M(1,2);
    const float S09 = M(1,0) * M(3,3) - M(3,0) * M(1,3);
    const float S10 = M(1,0) * M(3,2) - M(3,0) * M(1,2);
    const float S11 = M(1,1) * M(3,3) - M(3,1) * M(1,3);
    const float S12 = M(1,0) * M(3,1) - M(3,0) * M(1,1);
    const float S13 = M(1,2) * M(2,3) - M(2,2) * M(1,3);
    const float S14 = M(1,1) * M(2,3) - M(2,1) * M(1,3);
    const float S15 = M(1,1) * M(2,2) - M(2,1) * M(1,2);
    const float S16 = M(1,0) * M(2,3) - M(2,0) * M(1,3);
    const float S17 = M(1,0) * M(2,2) - M(2,0) * M(1,2);
    const float S18 = M(1,0) * M(2,1) - M(2,0) * M(1,1);
    float I[16];
    I[0]  = + ((M(1,1) * S00 - M(1,2) * S01) + M(1,3) * S02);
    I[1]  = - ((M(1,0) * S00 - M(1,2) * S03) + M(1,3) * S04);
    I[2]  = + ((M(1,0) * S01 - M(1,1) * S03) + M(1,3) * S05);
    I[3]  = - ((M(1,0) * S02 - M(1,1) * S04) + M(1,2) * S05);
    I[4]  = - ((M(0,1) * S00 - M(0,2) * S01) + M(0,3) * S02);
    I[5]  = + ((M(0,0) * S00 - M(0,2) * S03) + M(0,3) * S04);
    I[6]  = - ((M(0,0) * S01 - M(0,1) * S03) + M(0,3) * S05);
    I[7]  = + ((M(0,0) * S02 - M(0,1) * S04) + M(0,2) * S05);
    I[8]  = + ((M(0,1) * S06 - M(0,2) * S07) + M(0,3) * S08);
    I[9]  = - ((M(0,0) * S06 - M(0,2) * S09) + M(0,3) * S10);
    I[10] = + ((M(0,0) * S11 - M(0,1) * S09) + M(0,3) * S12);
    I[11] = - ((M(0,0) * S08 - M(0,1) * S10) + M(0,2) * S12);
    I[12] = - ((M(0,1) * S13 - M(0,2) * S14) + M(0,3) * S15);
    I[13] = + ((M(0,0) * S13 - M(0,2) * S16) + M(0,3) * S17);
    I[14] = - ((M(0,0) * S14 - M(0,1) * S16) + M(0,3) * S18);
    I[15] = + ((M(0,0) * S15 - M(0,1) * S17) + M(0,2) * S18);
    const float Det = ((+ M(0,0) * I[0] + M(0,1) * I[1]) + M(0,2) * I[2]) + M(0,3) * I[3];
    for (int i = 0; i < 16; i++) out[i] = I[i] / Det;
#undef M
}

static void mat_translate(const float* m, v3 v, float* r) {               /* gtc/matrix_transform.inl:40-50 */
    float t[16];
    memcpy(t, m, 64);
    for (int row = 0; row < 4; row++)
        t[12 + row] = ((m[0 + row] * v.x + m[4 + row] * v.y) + m[8 + row] * v.z) + m[12 + row];
    memcpy(r, t, 64);
}
static void mat_scale(const float* m, v3 v, float* r) {                   /* gtc/matrix_transform.inl:122-131 */
    float t[16];
    for (int row = 0; row < 4; row++) {
        t[0 + row] = m[0 + row] * v.x;
        t[4 + row] = m[4 + row] * v.y;
        t[8 + row] = m[8 + row] * v.z;
        t[12 + row] = m[12 + row];
    }
    memcpy(r, t, 64);
}

/* utilityCore::buildTransformationMatrix (utilities.cpp:45-52) + scene.cpp:92-95; fills the three matrices of g
 * from g->translation / rotation (degrees) / scale. */
void orc_build_geom(orc_geom* g) {
    float I[16], T[16], R[16], R2[16], S[16], TR[16];
    mat_identity(I);
    mat_translate(I, V(g->translation[0], g->translation[1], g->translation[2]), T);
    mat_rotate(I, g->rotation[0] * (float)PI_F / 180, V(1, 0, 0), R);
    mat_rotate(I, g->rotation[1] * (float)PI_F / 180, V(0, 1, 0), R2); mat_mul(R, R2, R);
    mat_rotate(I, g->rotation[2] * (float)PI_F / 180, V(0, 0, 1), R2); mat_mul(R, R2, R);
    mat_scale(I, V(g->scale[0], g->scale[1], g->scale[2]), S);
    mat_mul(T, R, TR);
    mat_mul(TR, S, g->transform);
    mat_inverse(g->transform, g->inverseTransform);
    mat_inverse_transpose(g->transform, g->invTranspose);
}

/* Scene::loadCamera (scene.cpp:143-152): fov/pixelLength/view from RES, FOVY, EYE, LOOKAT, UP.
 * cam->res, position, lookAt, up must be set; fovy in degrees (used as the HALF angle, as the reference does). */
void orc_camera_setup(orc_camera* cam, float fovy) {
    const float yscaled = tanf(fovy * (PI_F / 180));
    const float xscaled = (yscaled * cam->res[0]) / cam->res[1];
    const float fovx = (atanf(xscaled) * 180) / PI_F;
    cam->fov[0] = fovx; cam->fov[1] = fovy;
    cam->pixelLength[0] = 2 * xscaled / (float)cam->res[0];
    cam->pixelLength[1] = 2 * yscaled / (float)cam->res[1];
    const v3 view = vnormalize(vsub(V(cam->lookAt[0], cam->lookAt[1], cam->lookAt[2]),
                                    V(cam->position[0], cam->position[1], cam->position[2])));
    cam->view[0] = view.x; cam->view[1] = view.y; cam->view[2] = view.z;
    /* camera.right at :148 is computed from a not-yet-set view; runCuda() overwrites it before the first frame. */
    cam->right[0] = cam->right[1] = cam->right[2] = 0;
}

/* main() (main.cpp:66-78): derive the orbit parameters from the loaded camera. */
void orc_camera_orbit_params(const orc_camera* cam, float* zoom, float* phi, float* theta) {
    const v3 view = V(cam->view[0], cam->view[1], cam->view[2]);
    const v3 viewXZ = V(view.x, 0.0f, view.z);
    const v3 viewZY = V(0.0f, view.y, view.z);
    *phi = acosf(vdot(vnormalize(viewXZ), V(0, 0, -1)));
    *theta = acosf(vdot(vnormalize(viewZY), V(0, 1, 0)));
    *zoom = vlength(vsub(V(cam->position[0], cam->position[1], cam->position[2]),
                         V(cam->lookAt[0], cam->lookAt[1], cam->lookAt[2])));
}

/* runCuda() camera rebuild (main.cpp:122-140); right is NOT normalised (:133-135). */
void orc_camera_orbit(orc_camera* cam, float zoom, float phi, float theta) {
    v3 cp;
    cp.x = zoom * sinf(phi) * sinf(theta);
    cp.y = zoom * cosf(theta);
    cp.z = zoom * cosf(phi) * sinf(theta);
    const v3 v = vneg(vnormalize(cp));
    const v3 u = V(0, 1, 0);
    const v3 r = vcross(v, u);
    const v3 up = vcross(r, v);
    cam->view[0] = v.x; cam->view[1] = v.y; cam->view[2] = v.z;
    cam->up[0] = up.x; cam->up[1] = up.y; cam->up[2] = up.z;
    cam->right[0] = r.x; cam->right[1] = r.y; cam->right[2] = r.z;
    cam->position[0] = cp.x + cam->lookAt[0];
    cam->position[1] = cp.y + cam->lookAt[1];
    cam->position[2] = cp.z + cam->lookAt[2];
}

/* ---------------------------------------------------------------- known-answer hooks (tests/test_oracle_trace_kats.py)
 * One record in, one record out, in the layouts of oracle/ref_glm_kats.cpp (which evaluates the reference's vendored
 * GLM + utilities.cpp on the same inputs). */
void orc_kat_tri(const float* i, float* o) {
    v3 bary = V(0, 0, 0);
    const int hit = glm_intersect_ray_triangle(V(i[0], i[1], i[2]), V(i[3], i[4], i[5]), V(i[6], i[7], i[8]),
                                               V(i[9], i[10], i[11]), V(i[12], i[13], i[14]), &bary);
    o[0] = hit ? 1.0f : 0.0f; o[1] = bary.x; o[2] = bary.y; o[3] = bary.z;
}
void orc_kat_vec(const float* i, float* o) {
    const v3 a = V(i[0], i[1], i[2]), b = V(i[3], i[4], i[5]);
    const v3 c = vcross(a, b), n = vnormalize(a), r = vreflect(a, b), t = glm_refract(a, b, i[6]);
    o[0] = vdot(a, b);
    o[1] = c.x; o[2] = c.y; o[3] = c.z;
    o[4] = vlength(a);
    o[5] = n.x; o[6] = n.y; o[7] = n.z;
    o[8] = r.x; o[9] = r.y; o[10] = r.z;
    o[11] = t.x; o[12] = t.y; o[13] = t.z;
}
void orc_kat_mulmv(const float* i, float* o) {
    const float* m = i;
    const v3 r = mulMV(m, V(i[16], i[17], i[18]), i[19]);
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
    o[3] = (m[3] * i[16] + m[7] * i[17]) + (m[11] * i[18] + m[15] * i[19]);
}
void orc_kat_matmul(const float* i, float* o) { mat_mul(i, i + 16, o); }
void orc_kat_trs(const float* i, float* o) {
    orc_geom g;
    memset(&g, 0, sizeof(g));
    memcpy(g.translation, i, 12); memcpy(g.rotation, i + 3, 12); memcpy(g.scale, i + 6, 12);
    orc_build_geom(&g);
    memcpy(o, g.transform, 64); memcpy(o + 16, g.inverseTransform, 64); memcpy(o + 32, g.invTranspose, 64);
}
void orc_kat_xform(const float* i, float* o) {
    const v3 v = V(i[17], i[18], i[19]);
    mat_translate(i, v, o);
    mat_rotate(i, i[16], v, o + 16);
    mat_scale(i, v, o + 32);
}
void orc_kat_inverse(const float* i, float* o) { mat_inverse(i, o); mat_inverse_transpose(i, o + 16); }
void orc_kat_minmax(const float* i, float* o) { o[0] = glm_min(i[0], i[1]); o[1] = glm_max(i[0], i[1]); }
