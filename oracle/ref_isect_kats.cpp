// ref_isect_kats.cpp -- known-answer generator over the REFERENCE'S OWN intersections.h (test infrastructure).
//
// Compiles, where it lies under /root/reference, the reference's UNMODIFIED Inference/src/intersections.h (which includes
// its sceneStructs.h and utilities.h) with plain g++.  sceneStructs.h:5 includes <cuda_runtime.h>: the image carries a genuine
// NVIDIA copy of that header (the CUDA toolkit headers that ship inside triton's NVIDIA backend,
// <site-packages>/triton/backends/nvidia/include), located by oracle/Makefile; it defines __host__ / __device__ for a host
// compiler as NVIDIA defines them and is NOT a stand-in written for this build.  When that directory is absent the target is
// skipped and the committed fixtures stay the pin.  Recipe: oracle/Makefile target `ref` -> oracle/_ref/isect_kats
// (git-ignored, never committed).  Nothing of the reference is copied: this driver only CALLS the reference's functions on
// inputs it reads from stdin and writes the raw results to stdout; tests/golden/gen_trace_kats.py turns them into
// tests/golden/trace_isect_kats.npz, which pins oracle/trace_oracle.c (tests/test_oracle_trace_kats.py) bit for bit.
//
// What is called (paths relative to /root/reference/Inference/src):
//   utilhash                   intersections.h:12-20
//   boxIntersectionTest        intersections.h:52-94     (getPointOnRay :27-29, multiplyMV :34-36)
//   sphereIntersectionTest     intersections.h:106-148
//   triangleIntersectionTest   intersections.h:159-172   (incl. the F8 hit point and the interpolated normal)
//   RayAABBintersect           intersections.h:175-200
//   Geom matrices as scene.cpp:92-95 builds them: utilityCore::buildTransformationMatrix (utilities.cpp:45-52, compiled
//   where it lies) + glm::inverse + glm::inverseTranspose
//
// Two declarations this TU must supply because g++ is not nvcc (they are declarations in test code, not headers):
//   * unqualified min/max on floats (intersections.h:135,138,189-190): in CUDA device code these resolve to the toolkit's
//     global overloads `float min(float, float) { return fminf(a, b); }` (NVIDIA crt/math_functions.hpp:977-980 / :1105-1108, only
//     declared under nvcc) -- stated here with the same bodies;
//   * nothing else: sqrt / powf come from <cmath>.
//
// Usage: isect_kats <table> <n>   reads n fixed-size records from stdin, writes n records to stdout (layouts below).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static inline float min(const float a, const float b) { return fminf(a, b); }     // NVIDIA crt/math_functions.hpp:977-980
static inline float max(const float a, const float b) { return fmaxf(a, b); }     // NVIDIA crt/math_functions.hpp:1105-1108

#define GLM_FORCE_PURE
#include <glm/gtc/matrix_inverse.hpp>
#include "intersections.h"          // the reference's header, unmodified

static glm::vec3 v3(const float* p) { return glm::vec3(p[0], p[1], p[2]); }
static void put3(float* o, const glm::vec3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

static Geom make_geom(const float* trs, GeomType type) {      // scene.cpp:92-95
    Geom g;
    memset(&g, 0, sizeof g);
    g.type = type;
    g.translation = v3(trs); g.rotation = v3(trs + 3); g.scale = v3(trs + 6);
    g.transform = utilityCore::buildTransformationMatrix(g.translation, g.rotation, g.scale);
    g.inverseTransform = glm::inverse(g.transform);
    g.invTranspose = glm::inverseTranspose(g.transform);
    return g;
}

// in : trs[9] origin[3] direction[3]                       (15 floats)
// out: t, P[3], N[3], outside                              (8 floats; P/N/outside are 0 when t == -1: the reference leaves
//                                                           its out-parameters untouched on a miss)
static void t_prim(const float* i, float* o, GeomType type) {
    const Geom g = make_geom(i, type);
    Ray r; r.origin = v3(i + 9); r.direction = v3(i + 12);
    glm::vec3 P(0.0f), N(0.0f);
    bool outside = false;
    const float t = type == CUBE ? boxIntersectionTest(g, r, P, N, outside) : sphereIntersectionTest(g, r, P, N, outside);
    o[0] = t;
    if (t == -1.0f) { for (int k = 1; k < 8; ++k) o[k] = 0.0f; return; }
    put3(o + 1, P); put3(o + 4, N); o[7] = outside ? 1.0f : 0.0f;
}
static void t_box(const float* i, float* o) { t_prim(i, o, CUBE); }
static void t_sphere(const float* i, float* o) { t_prim(i, o, SPHERE); }

// in : origin[3] direction[3] v0 v1 v2 n0 n1 n2            (24 floats)
// out: t, P[3], N[3]                                       (7 floats; zeros after t on a miss)
static void t_tri_full(const float* i, float* o) {
    Face f;
    for (int k = 0; k < 3; ++k) { f.v[k] = v3(i + 6 + 3 * k); f.n[k] = v3(i + 15 + 3 * k); }
    f.materialid = 0;
    Ray r; r.origin = v3(i); r.direction = v3(i + 3);
    glm::vec3 P(0.0f), N(0.0f);
    bool outside = false;
    const float t = triangleIntersectionTest(f, r, P, N, outside);
    o[0] = t;
    if (t == -1.0f) { for (int k = 1; k < 7; ++k) o[k] = 0.0f; return; }
    put3(o + 1, P); put3(o + 4, N);
}

// in : origin[3] direction[3] lb[3] ub[3]                  (12 floats)   out: hit (1 float)
static void t_aabb(const float* i, float* o) {
    Ray r; r.origin = v3(i); r.direction = v3(i + 3);
    MeshBoundingBox b; b.lb = v3(i + 6); b.ub = v3(i + 9);
    o[0] = RayAABBintersect(r, b) ? 1.0f : 0.0f;
}

// in : one uint32 (as the bits of a float32 word)          out: utilhash (1 word)
static void t_utilhash(const float* i, float* o) {
    uint32_t a; memcpy(&a, i, 4);
    const uint32_t h = utilhash(a);
    memcpy(o, &h, 4);
}

struct Table { const char* name; int nin, nout; void (*fn)(const float*, float*); };
static const Table TABLES[] = {
    {"box", 15, 8, t_box}, {"sphere", 15, 8, t_sphere}, {"tri_full", 24, 7, t_tri_full}, {"aabb", 12, 1, t_aabb},
    {"utilhash", 1, 1, t_utilhash},
};

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s <table> <n>\n", argv[0]); return 2; }
    const int n = atoi(argv[2]);
    for (const Table& t : TABLES) {
        if (strcmp(t.name, argv[1])) continue;
        std::vector<float> in((size_t)n * t.nin), out((size_t)n * t.nout);
        if (fread(in.data(), 4, in.size(), stdin) != in.size()) { fprintf(stderr, "short input\n"); return 3; }
        for (int k = 0; k < n; ++k) t.fn(&in[(size_t)k * t.nin], &out[(size_t)k * t.nout]);
        fwrite(out.data(), 4, out.size(), stdout);
        return 0;
    }
    fprintf(stderr, "unknown table %s\n", argv[1]);
    return 2;
}
