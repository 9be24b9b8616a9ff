// internal.h -- context layout shared by the translation units of libaiptd.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/aiptd.h"

namespace aipt {

struct TraceState;     // trace.hip
struct DenoiseState;   // denoise.hip

}  // namespace aipt

// frames of one recurrent sequence whose denoiser passes are in flight together (independent sequences: 2 streams 1.25x, 3 streams 1.38x,
// 4 streams 1.37x the throughput of one, tools/overlap_probe.py; ONE sequence, level by level behind each other: 3 in flight = 2 in flight)
constexpr int AIPT_DN_PIPE = 2;
constexpr int AIPT_TRACE_BATCH_MAX = 24;    // frames one set of trace launches can hold (= BMAX of trace.hip)
constexpr int AIPT_TRACE_LANES_MIN = 2;     // aipt_frames: from this many frames on a trace call runs as two half-batches side by side
constexpr int AIPT_FRAMES_MAX = 32;         // frames one aipt_frames call can hold

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
    float* d_gbuf = nullptr;      // [10][fhp][fwp]: the G-buffer of the last aipt_frame (= d_gbufs[front])
    float* d_gbufs[2] = {nullptr, nullptr};
    int front = 0;
    // aipt_frame_prefetch: the next frame's trace runs on `st_trace` into the back G-buffer while this frame is denoised on
    // `st_dn` -- two streams restricted to DISJOINT sets of CUs (hipExtStreamCreateWithCUMask): a scheduling choice (a single
    // frame's trace does not fill the chip; sharing all CUs measured slower), no longer a correctness fence
    hipStream_t st_trace = nullptr, st_dn = nullptr;
    int st_dn_cus = 0;                                // CUs enabled on st_dn
    hipStream_t st_lane1 = nullptr;                   // aipt_frames: the second half of a call's frames is traced here, beside the first
    hipEvent_t ev_lane_fork = nullptr, ev_lane_join = nullptr;
    // aipt_frames: the denoiser passes of consecutive frames rotate over `stream` and the `pipe` streams (denoise_run,
    // pipelined): AIPT_DN_PIPE frames in flight
    hipStream_t pipe[AIPT_DN_PIPE - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[AIPT_DN_PIPE - 1] = {};
    hipEvent_t ev_denoised[2] = {nullptr, nullptr};   // last denoise that read d_gbufs[i] has finished
    hipEvent_t ev_prefetched = nullptr;               // the prefetched trace has finished
    hipEvent_t ev_traced = nullptr;                   // last trace (any stream) has finished: traces share one path state
    hipEvent_t ev_entry = nullptr;                    // aipt_frame: what the host had queued on `stream` before the call
    hipStream_t last_trace_stream = nullptr;
    bool denoised_valid[2] = {false, false};
    bool denoised_masked[2] = {false, false};         // that denoise ran on the CU-masked stream st_dn (a prefetched frame)
    struct { bool valid = false; aipt_camera cam; int iter = 0, depth = 0; uint32_t flags = 0; int buf = 0; } pf;
    // aipt_frames: a batch of frames traced together, then denoised in order
    int fbatch = 1;
    float* d_gbatch = nullptr;    // [fbatch][10][fhp][fwp]: the G-buffers of the batch being denoised (= d_gbatches[bfront])
    float* d_gbatches[2] = {nullptr, nullptr};   // double buffer: aipt_frames_prefetch traces the next batch into the back one
    int bfront = 0;
    hipEvent_t ev_bdenoised[2] = {nullptr, nullptr};   // the denoiser passes that read d_gbatches[i] have finished
    bool bdenoised_valid[2] = {false, false};
    struct { bool valid = false; std::vector<aipt_camera> cams; int iter = 0, depth = 0; uint32_t flags = 0; int buf = 0; } bpf;
    hipEvent_t bev[2] = {nullptr, nullptr};
    int last_batch = 1;
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

// 4-wide BVH node (bvh.cpp), 64 bytes = four 16-byte loads: quantisation frame (origin p, per-axis scale 2^(e-127)), the
// 8-bit child boxes (byte k of qlo[a] / qhi[a] = child k on axis a; decoded lo = fma(q, scale, p), always enclosing the padded
// fp32 box) and the child references: >= 0 inner node index, BVH_EMPTY = no child, otherwise a leaf
// -(first_leaf_face * 8 + count) - 1.  Root = node 0.
constexpr int BVH_LEAF_FACES = 7;    // (round 5: 4 -> 7, fewer dependent node visits for the same triangle tests: frame by frame 534 vs 521 frames/s, batched unchanged)
constexpr int BVH_EMPTY = INT32_MIN;
constexpr int BVH_MAX_STACK = 72;    // most stack entries per lane the traversal may need (LDS: entries x 1 KB per workgroup)
struct Bvh4Node {
    float p[3]; uint32_t exps;       // exps = ex | ey << 8 | ez << 16 | nchildren << 24
    uint32_t qlo[3], qhi[3];
    int ref[4];
    uint32_t pad[2];
};
static_assert(sizeof(Bvh4Node) == 64, "Bvh4Node");
// leaf triangle record, 48 bytes = three 16-byte loads: v0, e1 = v1 - v0, e2 = v2 - v0 (the fp32 subtractions
// glm::intersectRayTriangle starts with), the face's index in the caller's array
struct TriRec { float v0[3], e1[3], e2[3]; int face; int pad[2]; };
static_assert(sizeof(TriRec) == 48, "TriRec");
// returns the traversal-stack bound (entries per lane), -1: the mesh needs more than BVH_MAX_STACK
int build_bvh4(const aipt_face* faces, int nfaces, std::vector<Bvh4Node>& nodes, std::vector<int>& leaf_faces);

// aipt_trace on an explicit stream; orders itself after the previous trace when that ran on another stream
// (nframes > 1: a batch of frames traced by one set of launches, G-buffer f at d_gbuf + f * gbuf_frame floats)
int trace_on_stream(aipt_ctx* ctx, hipStream_t st, const aipt_camera* cam, int nframes, int iter, int depth, uint32_t flags,
                    float* d_gbuf, int gbuf_rows, int gbuf_stride, size_t gbuf_frame, int lane = 0);
// wait for the main stream and the denoiser pipeline streams
inline hipError_t sync_streams(aipt_ctx* ctx) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    for (hipStream_t ps : ctx->pipe)
        if (e == hipSuccess && ps) e = hipStreamSynchronize(ps);
    if (e == hipSuccess && ctx->st_trace && ctx->st_trace != ctx->stream) e = hipStreamSynchronize(ctx->st_trace);
    if (e == hipSuccess && ctx->st_dn) e = hipStreamSynchronize(ctx->st_dn);
    if (e == hipSuccess && ctx->st_lane1) e = hipStreamSynchronize(ctx->st_lane1);
    return e;
}
// aipt_denoise with the planar output cropped to out_h x out_w (<= the configured size)
int denoise_run(aipt_ctx* ctx, const float* d_in10, float* d_out3, uint32_t flags, int out_h, int out_w, bool pipelined = false,
                hipStream_t on = nullptr, int on_cus = 0);       // on: a CU-masked stream with on_cus CUs enabled
void trace_destroy(aipt_ctx* ctx);
int trace_enable_lanes(aipt_ctx* ctx);        // aipt_frames_configure: allocate the side lane of two-lane traces and warm its stream
bool trace_lanes_ready(aipt_ctx* ctx, int nframes);
bool trace_profiling(aipt_ctx* ctx);          // aipt_trace_profile_begin is recording: traces run one lane at a time
void denoise_destroy(aipt_ctx* ctx);

}  // namespace aipt
