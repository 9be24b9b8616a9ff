// abi.cpp -- context, error reporting, memory/timer helpers and the per-frame pipeline of libaiptd.so.
// Boundary: include/aiptd.h.  aipt_frame replaces the body of runCuda() (reference Inference/src/main.cpp:143-163):
// the G-buffer stays in HBM between the trace and the denoiser (the reference round-trips it through host memory,
// pathtrace.cu:525 + main.cpp:104-105).
#include "internal.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

namespace aipt {

static std::mutex g_err_mu;
static std::string g_err;

void set_global_error(const char* msg) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = msg;
}

int fail(aipt_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else set_global_error(buf);
    return code;
}

}  // namespace aipt

using aipt::fail;

extern "C" {

int aipt_abi_version(void) { return AIPT_ABI_VERSION; }

int aipt_create(int device, void* stream, aipt_ctx** out) {
    if (!out) return fail(nullptr, AIPT_E_INVALID, "aipt_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, AIPT_E_HIP, "aipt_create: no HIP device (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev) return fail(nullptr, AIPT_E_INVALID, "aipt_create: device %d of %d", device, ndev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, AIPT_E_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(nullptr, AIPT_E_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return fail(nullptr, AIPT_E_HIP, "aipt_create: device is %s; kernels are built for gfx950 only", prop.gcnArchName);
    aipt_ctx* ctx = new (std::nothrow) aipt_ctx();
    if (!ctx) return fail(nullptr, AIPT_E_NOMEM, "aipt_create: out of host memory");
    ctx->device = device;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return fail(nullptr, AIPT_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        ctx->own_stream = true;
    }
    hipEventCreate(&ctx->ev0);
    hipEventCreate(&ctx->ev1);
    for (auto& ev : ctx->fev) hipEventCreate(&ev);
    for (auto& ev : ctx->ev_denoised) hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_prefetched, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_traced, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_entry, hipEventDisableTiming);
    *out = ctx;
    return AIPT_OK;
}

void aipt_destroy(aipt_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (hipStream_t ps : ctx->pipe) if (ps) { hipStreamSynchronize(ps); hipStreamDestroy(ps); }
    for (hipStream_t ps : {ctx->st_trace, ctx->st_dn}) if (ps && ps != ctx->stream) { hipStreamSynchronize(ps); hipStreamDestroy(ps); }
    if (ctx->st_lane1) { hipStreamSynchronize(ctx->st_lane1); hipStreamDestroy(ctx->st_lane1); }
    if (ctx->ev_lane_fork) hipEventDestroy(ctx->ev_lane_fork);
    if (ctx->ev_lane_join) hipEventDestroy(ctx->ev_lane_join);
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    for (hipEvent_t ev : ctx->ev_join) if (ev) hipEventDestroy(ev);
    aipt::trace_destroy(ctx);
    aipt::denoise_destroy(ctx);
    for (float* g : ctx->d_gbufs) if (g) hipFree(g);
    for (float* g : ctx->d_gbatches) if (g) hipFree(g);
    for (auto& ev : ctx->bev) if (ev) hipEventDestroy(ev);
    for (auto& ev : ctx->ev_bdenoised) if (ev) hipEventDestroy(ev);
    for (auto& ev : ctx->ev_denoised) if (ev) hipEventDestroy(ev);
    if (ctx->ev_prefetched) hipEventDestroy(ctx->ev_prefetched);
    if (ctx->ev_traced) hipEventDestroy(ctx->ev_traced);
    if (ctx->ev_entry) hipEventDestroy(ctx->ev_entry);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    for (auto& ev : ctx->fev) if (ev) hipEventDestroy(ev);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* aipt_last_error(const aipt_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(aipt::g_err_mu);
    static thread_local std::string copy;
    copy = aipt::g_err;
    return copy.c_str();
}

int aipt_sync(aipt_ctx* ctx) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    return AIPT_OK;
}

int aipt_malloc(aipt_ctx* ctx, size_t bytes, void** d_out) {
    AIPT_CHECK_CTX(ctx);
    if (!d_out) return fail(ctx, AIPT_E_INVALID, "aipt_malloc: d_out is NULL");
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_out, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(ctx, AIPT_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return AIPT_OK;
}

int aipt_free(aipt_ctx* ctx, void* d_ptr) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    AIPT_HIP(ctx, hipFree(d_ptr));
    return AIPT_OK;
}

int aipt_upload(aipt_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    return AIPT_OK;
}

int aipt_download(aipt_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    return AIPT_OK;
}

int aipt_memset(aipt_ctx* ctx, void* d_dst, int value, size_t bytes) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return AIPT_OK;
}

int aipt_timer_start(aipt_ctx* ctx) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return AIPT_OK;
}

int aipt_timer_stop(aipt_ctx* ctx, float* ms_out) {
    AIPT_CHECK_CTX(ctx);
    AIPT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    AIPT_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    AIPT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (ms_out) *ms_out = ms;
    return AIPT_OK;
}

// ---------------------------------------------------------------------------------------------------- frame
static inline int round_up32(int v) { return (v + 31) & ~31; }

namespace aipt {
// the prefetch stream runs at the lowest priority: its trace workgroups fill what the denoiser leaves idle instead of
// delaying the convolutions of the frame being denoised (which are on the critical path of the sequence)
}  // namespace aipt

int aipt_frame_configure(aipt_ctx* ctx, int width, int height) {
    AIPT_CHECK_CTX(ctx);
    if (width <= 0 || height <= 0) return fail(ctx, AIPT_E_INVALID, "aipt_frame_configure: %dx%d", width, height);
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    const int wp = round_up32(width), hp = round_up32(height);   // pad policy: zeros bottom/right
    int rc = aipt_trace_configure(ctx, width, height);
    if (rc) return rc;
    rc = aipt_denoise_configure(ctx, hp, wp);
    if (rc) return rc;
    for (float*& g : ctx->d_gbufs) if (g) { hipFree(g); g = nullptr; }
    for (float*& g : ctx->d_gbatches) if (g) { hipFree(g); g = nullptr; }
    ctx->d_gbatch = nullptr; ctx->bpf.valid = false; ctx->bdenoised_valid[0] = ctx->bdenoised_valid[1] = false;
    ctx->fbatch = 1;
    ctx->d_gbuf = nullptr; ctx->front = 0; ctx->pf.valid = false;
    ctx->denoised_valid[0] = ctx->denoised_valid[1] = false;
    ctx->denoised_masked[0] = ctx->denoised_masked[1] = false;
    const size_t plane = (size_t)wp * hp;
    // two G-buffers: aipt_frame_prefetch traces the next frame into the back one while the front one is denoised.
    // A miss pixel is all-zero in the reference G-buffer; padding uses the same value and is never overwritten.
    for (float*& g : ctx->d_gbufs) {
        AIPT_HIP(ctx, hipMalloc((void**)&g, sizeof(float) * 10 * plane));
        AIPT_HIP(ctx, hipMemsetAsync(g, 0, sizeof(float) * 10 * plane, ctx->stream));
    }
    ctx->d_gbuf = ctx->d_gbufs[0];
    ctx->fw = width; ctx->fh = height; ctx->fwp = wp; ctx->fhp = hp;
    AIPT_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the zero fills are done before any stream traces
    return AIPT_OK;
}

int aipt_gbuffer(aipt_ctx* ctx, float** d_gbuf, int* rows, int* stride) {
    AIPT_CHECK_CTX(ctx);
    if (!ctx->d_gbuf) return fail(ctx, AIPT_E_STATE, "aipt_gbuffer: call aipt_frame_configure first");
    if (d_gbuf) *d_gbuf = ctx->d_gbuf;
    if (rows) *rows = ctx->fhp;
    if (stride) *stride = ctx->fwp;
    return AIPT_OK;
}

int aipt_frame_set_timing(aipt_ctx* ctx, int enabled) {
    AIPT_CHECK_CTX(ctx);
    ctx->frame_timing = enabled != 0;
    ctx->frame_timed = false;
    return AIPT_OK;
}

int aipt_frame(aipt_ctx* ctx, const aipt_camera* cam, int iter, int depth, uint32_t trace_flags, uint32_t dn_flags,
               float* d_out3) {
    AIPT_CHECK_CTX(ctx);
    if (!ctx->d_gbuf) return fail(ctx, AIPT_E_STATE, "aipt_frame: call aipt_frame_configure first");
    if (!cam || !d_out3) return fail(ctx, AIPT_E_INVALID, "aipt_frame: NULL argument");
    if (cam->resolution[0] != ctx->fw || cam->resolution[1] != ctx->fh)
        return fail(ctx, AIPT_E_INVALID, "aipt_frame: camera is %dx%d, configured %dx%d", cam->resolution[0],
                    cam->resolution[1], ctx->fw, ctx->fh);
    if (ctx->frame_timing) AIPT_HIP(ctx, hipEventRecord(ctx->fev[0], ctx->stream));
    int rc;
    const bool hit = ctx->pf.valid && ctx->pf.iter == iter && ctx->pf.depth == depth && ctx->pf.flags == trace_flags &&
                     !memcmp(&ctx->pf.cam, cam, sizeof(*cam));
    ctx->pf.valid = false;                                     // a mismatching prefetch is dropped (its G-buffer is reused)
    if (hit) {
        ctx->front = ctx->pf.buf;
        AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prefetched, 0));
    } else {                                                   // unpipelined: the front G-buffer stays where it is
        rc = aipt::trace_on_stream(ctx, ctx->stream, cam, 1, iter, depth, trace_flags, ctx->d_gbufs[ctx->front], ctx->fhp, ctx->fwp, 0);
        if (rc) return rc;
    }
    ctx->d_gbuf = ctx->d_gbufs[ctx->front];
    ctx->last_batch = 1;
    if (ctx->frame_timing) AIPT_HIP(ctx, hipEventRecord(ctx->fev[1], ctx->stream));
    // the final normalisation pass writes the cropped [3][h][w] image directly (no padded copy, no crop copies).
    // A prefetched frame is denoised on the CU-masked denoiser stream, so that the NEXT prefetch's trace (on the disjoint
    // CUs of st_trace) may run beside it; the pass follows the previous denoise (hidden state) and the prefetched trace.
    hipStream_t dn = hit && ctx->st_dn ? ctx->st_dn : ctx->stream;
    if (dn != ctx->stream) {
        // whatever the host queued on the context's stream before this call (an async read of the previous frame's d_out3 or
        // G-buffer, say) comes first
        AIPT_HIP(ctx, hipEventRecord(ctx->ev_entry, ctx->stream));
        AIPT_HIP(ctx, hipStreamWaitEvent(dn, ctx->ev_entry, 0));
        AIPT_HIP(ctx, hipStreamWaitEvent(dn, ctx->ev_prefetched, 0));
        for (int b = 0; b < 2; b++) if (ctx->denoised_valid[b]) AIPT_HIP(ctx, hipStreamWaitEvent(dn, ctx->ev_denoised[b], 0));
    }
    rc = aipt::denoise_run(ctx, ctx->d_gbuf, d_out3, dn_flags, ctx->fh, ctx->fw, false, dn != ctx->stream ? dn : nullptr, ctx->st_dn_cus);
    if (rc) return rc;
    AIPT_HIP(ctx, hipEventRecord(ctx->ev_denoised[ctx->front], dn));
    ctx->denoised_valid[ctx->front] = true;
    ctx->denoised_masked[ctx->front] = dn != ctx->stream;
    if (dn != ctx->stream) AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_denoised[ctx->front], 0));   // join
    if (ctx->frame_timing) {
        AIPT_HIP(ctx, hipEventRecord(ctx->fev[2], ctx->stream));
        ctx->frame_timed = true;
    }
    return AIPT_OK;
}

// ---- batches of frames: one set of trace launches for n consecutive frames, then n denoiser passes in order
int aipt_frames_configure(aipt_ctx* ctx, int batch) {
    AIPT_CHECK_CTX(ctx);
    if (!ctx->d_gbuf) return fail(ctx, AIPT_E_STATE, "aipt_frames_configure: call aipt_frame_configure first");
    if (batch < 1 || batch > AIPT_FRAMES_MAX) return fail(ctx, AIPT_E_INVALID, "aipt_frames_configure: batch %d (1..%d)", batch, AIPT_FRAMES_MAX);
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    AIPT_HIP(ctx, aipt::sync_streams(ctx));
    int rc = aipt_trace_configure_batch(ctx, ctx->fw, ctx->fh, batch < AIPT_TRACE_BATCH_MAX ? batch : AIPT_TRACE_BATCH_MAX);
    if (!rc) rc = aipt::trace_enable_lanes(ctx);
    if (rc) return rc;
    for (float*& g : ctx->d_gbatches) if (g) { hipFree(g); g = nullptr; }
    const size_t frame = (size_t)10 * ctx->fwp * ctx->fhp;
    for (float*& g : ctx->d_gbatches) {
        AIPT_HIP(ctx, hipMalloc((void**)&g, sizeof(float) * frame * batch));
        AIPT_HIP(ctx, hipMemsetAsync(g, 0, sizeof(float) * frame * batch, ctx->stream));   // padding = miss pixels = 0
    }
    AIPT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->bev[0]) for (auto& ev : ctx->bev) hipEventCreate(&ev);
    if (!ctx->ev_bdenoised[0]) for (auto& ev : ctx->ev_bdenoised) hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    ctx->bfront = 0; ctx->d_gbatch = ctx->d_gbatches[0];
    ctx->bdenoised_valid[0] = ctx->bdenoised_valid[1] = false; ctx->bpf.valid = false;
    ctx->fbatch = batch;
    ctx->pf.valid = false;
    return AIPT_OK;
}

// the traces of a batch of frames: one set of launches per AIPT_TRACE_BATCH_MAX frames (the trace's cameras travel as kernel
// arguments), one after the other on `st`, into consecutive G-buffers
static int trace_frames(aipt_ctx* ctx, hipStream_t st, const aipt_camera* cams, int nframes, int iter, int depth, uint32_t trace_flags,
                        float* d_gbatch) {
    const size_t frame = (size_t)10 * ctx->fwp * ctx->fhp;
    // as few calls as AIPT_TRACE_BATCH_MAX allows, of (nearly) equal size: 20 frames are traced 10 + 10, not 16 + 4 (a
    // 4-frame call's launches are mostly tail)
    const int ncalls = (nframes + AIPT_TRACE_BATCH_MAX - 1) / AIPT_TRACE_BATCH_MAX;
    // [r5] Two lanes: from AIPT_TRACE_LANES_MIN (2) frames on, the two halves of a call's frames are traced BESIDE each other -- the
    // first on `st`, the second on a side stream with its own path state.  Every bounce launch ends in a tail (its last workgroups
    // walk their longest rays while the rest of the chip has run out of work: a single frame's later bounces last 105-150 us
    // whatever their ray count), seven in a row per trace; two launch sequences side by side fill each other's tails: 0.333 vs
    // 0.382 ms per frame at 20 frames, 0.326 vs 0.387 at 24 (tools/dual_trace_probe.py); calls of 2 / 4 / 8 frames: 718 / 813 / 887
    // against 668 / 756 / 804 frames/s.  Frames are independent, so the split changes no bit (tests/test_gpu_frame.py).
    // AIPT_TRACE_LANES=1 keeps one lane (scheduling only; profiling passes that want one kernel at a time).
    static const int lanes_env = getenv("AIPT_TRACE_LANES") ? atoi(getenv("AIPT_TRACE_LANES")) : 2;
    for (int c = 0, k = 0; c < ncalls; c++) {
        const int nb = nframes / ncalls + (c < nframes % ncalls ? 1 : 0);
        // (one lane while aipt_trace_profile_* records: an event pair then brackets a launch that has the chip to itself)
        const int na_two = (nb + 1) / 2;
        const bool two = lanes_env >= 2 && nb >= AIPT_TRACE_LANES_MIN && iter == 1 && !aipt::trace_profiling(ctx) &&
                         aipt::trace_lanes_ready(ctx, nb - na_two);
        const int na = two ? na_two : nb;
        if (two) {
            // the side lane starts where `st` stands now (the G-buffers it writes were last read by work queued on `st`) ...
            AIPT_HIP(ctx, hipEventRecord(ctx->ev_lane_fork, st));
            AIPT_HIP(ctx, hipStreamWaitEvent(ctx->st_lane1, ctx->ev_lane_fork, 0));
        }
        int rc = aipt::trace_on_stream(ctx, st, cams + k, na, iter, depth, trace_flags, d_gbatch + k * frame, ctx->fhp, ctx->fwp, frame);
        if (rc) return rc;
        if (two) {
            rc = aipt::trace_on_stream(ctx, ctx->st_lane1, cams + k + na, nb - na, iter, depth, trace_flags, d_gbatch + (k + na) * frame,
                                       ctx->fhp, ctx->fwp, frame, 1);
            if (rc) return rc;
            // ... and `st` goes on when both halves are done
            AIPT_HIP(ctx, hipEventRecord(ctx->ev_lane_join, ctx->st_lane1));
            AIPT_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_lane_join, 0));
        }
        k += nb;
    }
    return AIPT_OK;
}

int aipt_frames(aipt_ctx* ctx, const aipt_camera* cams, int nframes, int iter, int depth, uint32_t trace_flags,
                uint32_t dn_flags_first, uint32_t dn_flags_rest, float* const* d_out3) {
    AIPT_CHECK_CTX(ctx);
    if (!ctx->d_gbatch) return fail(ctx, AIPT_E_STATE, "aipt_frames: call aipt_frames_configure first");
    if (!cams || !d_out3 || nframes < 1 || nframes > ctx->fbatch)
        return fail(ctx, AIPT_E_INVALID, "aipt_frames: %d frames, configured for %d", nframes, ctx->fbatch);
    for (int j = 0; j < nframes; j++) if (!d_out3[j]) return fail(ctx, AIPT_E_INVALID, "aipt_frames: output %d is NULL", j);
    ctx->pf.valid = false;
    const size_t frame = (size_t)10 * ctx->fwp * ctx->fhp;
    if (ctx->frame_timing) AIPT_HIP(ctx, hipEventRecord(ctx->fev[0], ctx->stream));
    int rc;
    const bool hit = ctx->bpf.valid && ctx->bpf.iter == iter && ctx->bpf.depth == depth && ctx->bpf.flags == trace_flags &&
                     (int)ctx->bpf.cams.size() == nframes && !memcmp(ctx->bpf.cams.data(), cams, sizeof(aipt_camera) * nframes);
    ctx->bpf.valid = false;                                    // a mismatching prefetch is dropped
    if (hit) {
        ctx->bfront = ctx->bpf.buf;
        AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prefetched, 0));
    } else {
        rc = trace_frames(ctx, ctx->stream, cams, nframes, iter, depth, trace_flags, ctx->d_gbatches[ctx->bfront]);
        if (rc) return rc;
    }
    ctx->d_gbatch = ctx->d_gbatches[ctx->bfront];
    if (ctx->frame_timing) AIPT_HIP(ctx, hipEventRecord(ctx->fev[1], ctx->stream));
    // the frames' denoiser passes rotate over AIPT_DN_PIPE streams, each frame following the one before it level by level
    // (denoise_run): fork after the trace, join before anything that follows on the context's stream
    static const bool pipe_env = !getenv("AIPT_DN_PIPELINE") || atoi(getenv("AIPT_DN_PIPELINE")) != 0;
    const bool pipelined = pipe_env && nframes > 1;
    if (pipelined) {
        if (!ctx->ev_fork) {
            for (int k = 0; k < AIPT_DN_PIPE - 1; k++) {
                AIPT_HIP(ctx, hipStreamCreateWithFlags(&ctx->pipe[k], hipStreamNonBlocking));
                AIPT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming));
            }
            AIPT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        }
        AIPT_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        for (hipStream_t ps : ctx->pipe) AIPT_HIP(ctx, hipStreamWaitEvent(ps, ctx->ev_fork, 0));
    }
    for (int j = 0; j < nframes; j++) {
        rc = aipt::denoise_run(ctx, ctx->d_gbatch + j * frame, d_out3[j], j == 0 ? dn_flags_first : dn_flags_rest, ctx->fh, ctx->fw, pipelined);
        if (rc) return rc;
    }
    if (pipelined) {
        for (int k = 0; k < AIPT_DN_PIPE - 1; k++) {
            AIPT_HIP(ctx, hipEventRecord(ctx->ev_join[k], ctx->pipe[k]));
            AIPT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[k], 0));
        }
    }
    AIPT_HIP(ctx, hipEventRecord(ctx->ev_bdenoised[ctx->bfront], ctx->stream));
    ctx->bdenoised_valid[ctx->bfront] = true;
    ctx->d_gbuf = ctx->d_gbatch + (size_t)(nframes - 1) * frame;
    ctx->last_batch = nframes;
    if (ctx->frame_timing) {
        AIPT_HIP(ctx, hipEventRecord(ctx->fev[2], ctx->stream));
        ctx->frame_timed = true;
    }
    return AIPT_OK;
}

int aipt_frames_prefetch(aipt_ctx* ctx, const aipt_camera* cams, int nframes, int iter, int depth, uint32_t trace_flags) {
    AIPT_CHECK_CTX(ctx);
    if (!ctx->d_gbatch) return fail(ctx, AIPT_E_STATE, "aipt_frames_prefetch: call aipt_frames_configure first");
    if (!cams || nframes < 1 || nframes > ctx->fbatch) return fail(ctx, AIPT_E_INVALID, "aipt_frames_prefetch: %d frames, configured for %d", nframes, ctx->fbatch);
    if (iter != 1) return fail(ctx, AIPT_E_INVALID, "aipt_frames_prefetch: only iter == 1 frames can be prefetched (iter %d)", iter);
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    const int back = ctx->bfront ^ 1;
    // On the context's stream, behind the denoiser passes already queued: a batch's bounce launches fill the chip, and beside the
    // persistent conv3x3_f16x3r launches (which need every CU whole) the two starve each other -- 625 vs 797 frames/s with the
    // trace on a low-priority side stream (tools/experiments/README.md, round 4).  The call still moves the trace
    // ahead of the host's next aipt_frames and keeps the double buffer.
    const int rc = trace_frames(ctx, ctx->stream, cams, nframes, iter, depth, trace_flags, ctx->d_gbatches[back]);
    if (rc) return rc;
    AIPT_HIP(ctx, hipEventRecord(ctx->ev_prefetched, ctx->stream));
    ctx->bpf.valid = true; ctx->bpf.cams.assign(cams, cams + nframes); ctx->bpf.iter = iter; ctx->bpf.depth = depth;
    ctx->bpf.flags = trace_flags; ctx->bpf.buf = back;
    return AIPT_OK;
}

int aipt_frames_gbuffer(aipt_ctx* ctx, int frame, float** d_gbuf, int* rows, int* stride) {
    AIPT_CHECK_CTX(ctx);
    if (!ctx->d_gbatch || frame < 0 || frame >= ctx->fbatch) return fail(ctx, AIPT_E_STATE, "aipt_frames_gbuffer: frame %d of %d", frame, ctx->fbatch);
    if (d_gbuf) *d_gbuf = ctx->d_gbatch + (size_t)frame * 10 * ctx->fwp * ctx->fhp;
    if (rows) *rows = ctx->fhp;
    if (stride) *stride = ctx->fwp;
    return AIPT_OK;
}

int aipt_frame_prefetch(aipt_ctx* ctx, const aipt_camera* cam, int iter, int depth, uint32_t trace_flags) {
    AIPT_CHECK_CTX(ctx);
    if (!ctx->d_gbuf) return fail(ctx, AIPT_E_STATE, "aipt_frame_prefetch: call aipt_frame_configure first");
    if (!cam) return fail(ctx, AIPT_E_INVALID, "aipt_frame_prefetch: NULL camera");
    if (cam->resolution[0] != ctx->fw || cam->resolution[1] != ctx->fh)
        return fail(ctx, AIPT_E_INVALID, "aipt_frame_prefetch: camera is %dx%d, configured %dx%d", cam->resolution[0],
                    cam->resolution[1], ctx->fw, ctx->fh);
    // planes 3-9 are written at iter == 1 only and live in the G-buffer the iteration-1 trace wrote (pathtrace.cu:295,379): a
    // later iteration traced into the OTHER buffer would be denoised with stale planes, so only iteration 1 can be prefetched
    if (iter != 1) return fail(ctx, AIPT_E_INVALID, "aipt_frame_prefetch: only iter == 1 frames can be prefetched (iter %d)", iter);
    AIPT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->st_trace) {
        // two streams on disjoint CUs: the first AIPT_PREFETCH_TRACE_CUS for the trace, the rest for the denoiser.  Mask bit i is one
        // CU of XCD i % 8, and a launch's workgroup b still starts on XCD b & 7 (tools/ubench/cumask.hip), so both shares are whole
        // multiples of 8 -- of 32, in fact: the persistent conv kernel takes one workgroup per enabled CU (denoise_run's on_cus),
        // and with a CU count per XCD that is not a multiple of 4 (its shader engines) some of them wait for a second round
        // (measured: trace share 64 / 72 / 80 / 88 / 96 / 104 / 112 / 120 / 128 CUs = 631 / 639 / 625 / 661 / 808 / 601 / 562 / 596 /
        // 719 frames/s).  Default 3/8 of the chip [r5: the split walks made the trace the shorter stage; rounds 2-4: half]
        hipDeviceProp_t prop;
        AIPT_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        const int ncu = prop.multiProcessorCount;
        static const int env_cus = getenv("AIPT_PREFETCH_TRACE_CUS") ? atoi(getenv("AIPT_PREFETCH_TRACE_CUS")) : 0;
        const int want = env_cus >= 8 && env_cus <= ncu - 32 ? env_cus : ncu * 3 / 8;
        // (the DENOISER's share in whole multiples of 32 = 4 CUs per XCD; the trace takes the rest, a multiple of 8 on a chip of 8 k CUs)
        const int dn = std::min(ncu - 8, std::max(32, (ncu - want + 16) / 32 * 32));
        const int nt = ncu >= 64 && (ncu & 7) == 0 ? ncu - dn : ncu / 2;
        std::vector<uint32_t> mt((ncu + 31) / 32, 0u), md((ncu + 31) / 32, 0u);
        for (int c = 0; c < ncu; c++) (c < nt ? mt : md)[c / 32] |= 1u << (c % 32);
        if (hipExtStreamCreateWithCUMask(&ctx->st_trace, (uint32_t)mt.size(), mt.data()) != hipSuccess ||
            hipExtStreamCreateWithCUMask(&ctx->st_dn, (uint32_t)md.size(), md.data()) != hipSuccess) {
            // no CU masks on this runtime: the trace is queued on the context's stream, behind the denoise (never beside it)
            (void)hipGetLastError();
            if (ctx->st_trace) hipStreamDestroy(ctx->st_trace);
            ctx->st_trace = ctx->stream; ctx->st_dn = nullptr;
        } else ctx->st_dn_cus = ncu - nt;
    }
    const int back = ctx->front ^ 1;
    hipStream_t ts = ctx->st_trace;
    // the trace follows everything queued on the context's stream so far except the denoise it is meant to overlap: the
    // denoise that last read the back G-buffer (trace_on_stream orders it behind the previous trace by itself)
    if (ts != ctx->stream && ctx->denoised_valid[back]) AIPT_HIP(ctx, hipStreamWaitEvent(ts, ctx->ev_denoised[back], 0));
    // ... and, when the frame being denoised was NOT prefetched (the first frame of a sequence, a changed request), its
    // denoise runs on the context's stream without a CU mask: the trace may only overlap a denoise on the masked stream
    if (ts != ctx->stream && ctx->denoised_valid[ctx->front] && !ctx->denoised_masked[ctx->front])
        AIPT_HIP(ctx, hipStreamWaitEvent(ts, ctx->ev_denoised[ctx->front], 0));
    const int rc = aipt::trace_on_stream(ctx, ts, cam, 1, iter, depth, trace_flags, ctx->d_gbufs[back], ctx->fhp, ctx->fwp, 0);
    if (rc) return rc;
    AIPT_HIP(ctx, hipEventRecord(ctx->ev_prefetched, ts));
    ctx->pf.valid = true; ctx->pf.cam = *cam; ctx->pf.iter = iter; ctx->pf.depth = depth; ctx->pf.flags = trace_flags;
    ctx->pf.buf = back;
    return AIPT_OK;
}

int aipt_frame_last_times(aipt_ctx* ctx, float* trace_ms, float* denoise_ms) {
    AIPT_CHECK_CTX(ctx);
    if (!ctx->frame_timed) return fail(ctx, AIPT_E_STATE, "aipt_frame_last_times: no timed frame");
    AIPT_HIP(ctx, hipEventSynchronize(ctx->fev[2]));
    float a = 0, b = 0;
    AIPT_HIP(ctx, hipEventElapsedTime(&a, ctx->fev[0], ctx->fev[1]));
    AIPT_HIP(ctx, hipEventElapsedTime(&b, ctx->fev[1], ctx->fev[2]));
    // after aipt_frames: per-frame averages over the batch
    if (trace_ms) *trace_ms = a / (float)ctx->last_batch;
    if (denoise_ms) *denoise_ms = b / (float)ctx->last_batch;
    return AIPT_OK;
}

}  // extern "C"
