// internal.h -- context layout shared by the translation units of libaiptd.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/aiptd.h"

namespace aipt {

struct TraceState;     // trace.hip
struct DenoiseState;   // denoise.hip

}  // namespace aipt

struct aipt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;          // aipt_timer_*
    hipEvent_t fev[3] = {nullptr, nullptr, nullptr};  // aipt_frame stage timing
    bool frame_timing = false;
    bool frame_timed = false;
    aipt::TraceState* trace = nullptr;
    aipt::DenoiseState* dn = nullptr;
    // aipt_frame
    int fw = 0, fh = 0, fwp = 0, fhp = 0;
    float* d_gbuf = nullptr;      // [10][fhp][fwp]
    float* d_out_pad = nullptr;   // [3][fhp][fwp] when cropping is needed
};

namespace aipt {

int fail(aipt_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
void set_global_error(const char* msg);

#define AIPT_HIP(ctx, expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return aipt::fail((ctx), AIPT_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define AIPT_CHECK_CTX(ctx) \
    do { if (!(ctx)) return AIPT_E_INVALID; } while (0)

// threaded BVH node (bvh.cpp): DFS order; leaf = (first_leaf_face << 3) | count, or -1 for an inner node
struct BvhNode {
    float lo[3];
    int skip;
    float hi[3];
    int leaf;
};
void build_bvh(const aipt_face* faces, int nfaces, std::vector<BvhNode>& nodes, std::vector<int>& leaf_faces);

void trace_destroy(aipt_ctx* ctx);
void denoise_destroy(aipt_ctx* ctx);

}  // namespace aipt
