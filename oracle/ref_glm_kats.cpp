// ref_glm_kats.cpp -- known-answer generator over the REFERENCE'S OWN third-party/host code (test infrastructure).
//
// Compiles, where they lie under /root/reference, (a) the reference's vendored GLM 0.9.6.3 headers
// (Inference/external/include/glm -- plain C++ headers, no CUDA needed) and (b) the reference's own host source file
// Inference/src/utilities.cpp (utilityCore::buildTransformationMatrix, :45-52; it includes only GLM and the standard
// library).  Recipe: oracle/Makefile target `ref` -> oracle/_ref/glm_kats (git-ignored, never committed).  Nothing of
// the reference is copied into this repository: this driver only CALLS the reference code on inputs it reads from
// stdin and writes the raw results to stdout; tests/golden/gen_trace_kats.py turns them into tests/golden/trace_glm_kats.npz,
// which pins oracle/trace_oracle.c (tests/test_oracle_trace_kats.py) bit for bit.
//
// What the reference calls and where (paths relative to /root/reference/Inference/src):
//   glm::intersectRayTriangle            intersections.h:164           (gtx/intersect.inl:37-74)
//   glm::dot/cross/length/normalize      intersections.h, interactions.h, pathtrace.cu:172-175, main.cpp:66-78,122-140
//   glm::reflect / glm::refract          interactions.h:226,233,236    (detail/func_geometric.inl:175-200)
//   glm::min / glm::max                  intersections.h:64-65         (detail/func_common.inl:407-435)
//   mat4 * vec4 (multiplyMV)             intersections.h:34-36         (detail/type_mat4x4.inl)
//   glm::translate/rotate/scale, mat4*mat4   utilities.cpp:45-52      (gtc/matrix_transform.inl)
//   glm::inverse / glm::inverseTranspose scene.cpp:92-95               (detail/func_matrix.inl, gtc/matrix_inverse.inl)
//
// Usage: glm_kats <table> <n>   reads n records of float32 from stdin, writes n records of float32 to stdout.
#define GLM_FORCE_PURE
#include <glm/glm.hpp>
#include <glm/gtc/matrix_inverse.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtx/intersect.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "utilities.h"      // the reference's header: utilityCore::buildTransformationMatrix

static glm::vec3 v3(const float* p) { return glm::vec3(p[0], p[1], p[2]); }
static void put3(float* o, const glm::vec3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
static glm::mat4 m4(const float* p) { glm::mat4 m; memcpy(&m[0][0], p, 64); return m; }     // column-major, as GLM stores it
static void putm(float* o, const glm::mat4& m) { memcpy(o, &m[0][0], 64); }

struct Table { const char* name; int nin, nout; void (*fn)(const float*, float*); };

static void t_tri(const float* i, float* o) {
    glm::vec3 bary(0.0f);
    const bool hit = glm::intersectRayTriangle(v3(i), v3(i + 3), v3(i + 6), v3(i + 9), v3(i + 12), bary);
    o[0] = hit ? 1.0f : 0.0f;
    put3(o + 1, bary);          // components computed before an early return are still written (intersect.inl:58-71)
}
static void t_vec(const float* i, float* o) {
    const glm::vec3 a = v3(i), b = v3(i + 3);
    const float eta = i[6];
    o[0] = glm::dot(a, b);
    put3(o + 1, glm::cross(a, b));
    o[4] = glm::length(a);
    put3(o + 5, glm::normalize(a));
    put3(o + 8, glm::reflect(a, b));
    put3(o + 11, glm::refract(a, b, eta));
}
static void t_mulmv(const float* i, float* o) {
    const glm::vec4 r = m4(i) * glm::vec4(i[16], i[17], i[18], i[19]);
    o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
}
static void t_matmul(const float* i, float* o) { putm(o, m4(i) * m4(i + 16)); }
static void t_trs(const float* i, float* o) {
    const glm::mat4 t = utilityCore::buildTransformationMatrix(v3(i), v3(i + 3), v3(i + 6));
    putm(o, t);
    putm(o + 16, glm::inverse(t));
    putm(o + 32, glm::inverseTranspose(t));
}
static void t_xform(const float* i, float* o) {
    const glm::mat4 m = m4(i);
    putm(o, glm::translate(m, v3(i + 17)));
    putm(o + 16, glm::rotate(m, i[16], v3(i + 17)));
    putm(o + 32, glm::scale(m, v3(i + 17)));
}
static void t_inverse(const float* i, float* o) {
    putm(o, glm::inverse(m4(i)));
    putm(o + 16, glm::inverseTranspose(m4(i)));
}
static void t_minmax(const float* i, float* o) {
    o[0] = glm::min(i[0], i[1]);
    o[1] = glm::max(i[0], i[1]);
}

static const Table TABLES[] = {
    {"tri", 15, 4, t_tri},       {"vec", 7, 14, t_vec},         {"mulmv", 20, 4, t_mulmv}, {"matmul", 32, 16, t_matmul},
    {"trs", 9, 48, t_trs},       {"xform", 20, 48, t_xform},    {"inverse", 16, 32, t_inverse}, {"minmax", 2, 2, t_minmax},
};

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: glm_kats <table> <n>\n"); return 2; }
    const int n = atoi(argv[2]);
    for (const Table& t : TABLES) {
        if (strcmp(t.name, argv[1])) continue;
        std::vector<float> in((size_t)n * t.nin), out((size_t)n * t.nout);
        if (fread(in.data(), 4, in.size(), stdin) != in.size()) { fprintf(stderr, "short input\n"); return 1; }
        for (int k = 0; k < n; k++) t.fn(&in[(size_t)k * t.nin], &out[(size_t)k * t.nout]);
        fwrite(out.data(), 4, out.size(), stdout);
        return 0;
    }
    fprintf(stderr, "unknown table %s\n", argv[1]);
    return 2;
}
