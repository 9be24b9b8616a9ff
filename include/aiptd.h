/*
 * aiptd.h -- C ABI of the MI355X-native 1-spp path-trace + recurrent-denoise hot path.
 *
 * Drop-in boundary for the reference's hot path (paths relative to /root/reference/Inference):
 *
 *   reference interface                                            replaced by
 *   -------------------------------------------------------------  -----------------------------------------
 *   void pathtraceInit(Scene*)              src/pathtrace.h:6      aipt_scene_upload + aipt_trace_configure
 *   void pathtraceFree()                    src/pathtrace.h:7      aipt_scene_free / aipt_destroy
 *   void pathtrace(uchar4*, int, int)       src/pathtrace.h:8      aipt_trace
 *   scene->state.host_tensor (float[10*W*H]) src/sceneStructs.h:74 the device G-buffer (aipt_gbuffer) /
 *        filled by cudaMemcpy D2H           src/pathtrace.cu:525   aipt_download when a host copy is wanted
 *   torch::jit::load(MODEL_PATH)            src/main.cpp:107       aipt_denoise_load_weights
 *   module.forward({[1,10,W,H]}) -> [1,3,W,H] src/main.cpp:104-111 aipt_denoise
 *   runCuda() per-frame body                src/main.cpp:143-163   aipt_frame
 *   Scene::Scene(filename)                  src/scene.cpp:11-320   aipt_scene_load (host-side front end)
 *   camera orbit rebuild in runCuda()       src/main.cpp:122-140   aipt_camera_orbit
 *
 * The POD structs below have the byte layout of the reference's sceneStructs.h types (glm::vec3 = 3 floats,
 * glm::mat4 = 16 floats column-major), so a reference-side caller can pass scene->geoms.data() etc. directly
 * (INTEGRATION.md shows the binding).
 *
 * Conventions: every function returns AIPT_OK (0) or a negative AIPT_E_* code; aipt_last_error() gives the message.
 * No C++ exception crosses this boundary.  One context per GPU and per host thread; all work is enqueued on the
 * context's HIP stream and is asynchronous unless stated otherwise.  Pointers named d_* are device pointers.
 * There is no CPU fallback: without a usable HIP device aipt_create fails.
 *
 * Environment variables the release library reads -- SCHEDULING only: each changes which launches share the chip or how many
 * frames one set of launches covers, never a bit of any result (tests/test_gpu_frame.py, tests/test_cli.py compare them):
 *   AIPT_DN_PIPELINE=0          aipt_frames: the denoiser passes of a call on ONE stream instead of two (profiling: a chip-wide
 *                               counter then belongs to one kernel; tools/collect_evidence.sh)
 *   AIPT_PREFETCH_TRACE_CUS=n   aipt_frame_prefetch: CUs of the trace stream's mask, a multiple of 32 (default: 3/8 of the chip)
 *   AIPT_TRACE_LANES=1          aipt_frames: trace a call's frames with ONE set of launches instead of two half-batches side by
 *                               side on two streams (default 2)
 * Everything else (kernel selection, tile sizes) is an explicit ABI option (aipt_denoise_set_option) or a debug-build hook
 * behind -DAIPT_DEBUG_HOOKS; tests/test_shipped_kernels_cpu.py fails on any other getenv in the kernels' sources.
 */
#ifndef AIPTD_H
#define AIPTD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIPT_ABI_VERSION 1

#define AIPT_OK          0
#define AIPT_E_INVALID  -1   /* bad argument */
#define AIPT_E_HIP      -2   /* HIP runtime error */
#define AIPT_E_STATE    -3   /* call out of order (e.g. trace before scene upload) */
#define AIPT_E_NOMEM    -4
#define AIPT_E_FORMAT   -5   /* malformed weight blob / scene file */
#define AIPT_E_IO       -6

typedef struct aipt_ctx aipt_ctx;

/* sceneStructs.h:8-13 */
#define AIPT_GEOM_SPHERE 0
#define AIPT_GEOM_CUBE   1

typedef struct aipt_geom {            /* Geom, sceneStructs.h:20-30, 248 bytes */
    int type;
    int materialid;
    float translation[3];
    float rotation[3];
    float scale[3];
    float transform[16];              /* column-major */
    float inverseTransform[16];
    float invTranspose[16];
    float vel[3];
} aipt_geom;

typedef struct aipt_face {            /* Face, sceneStructs.h:40-44, 76 bytes */
    float v[3][3];
    float n[3][3];
    int materialid;
} aipt_face;

typedef struct aipt_material {        /* Material, sceneStructs.h:46-56, 44 bytes */
    float color[3];
    float specular_exponent;
    float specular_color[3];
    float hasReflective;
    float hasRefractive;
    float indexOfRefraction;
    float emittance;
} aipt_material;

typedef struct aipt_camera {          /* Camera, sceneStructs.h:58-67, 84 bytes */
    int resolution[2];                /* x = width, y = height */
    float position[3];
    float lookAt[3];
    float view[3];
    float up[3];
    float right[3];
    float fov[2];
    float pixelLength[2];
} aipt_camera;

typedef struct aipt_aabb {            /* MeshBoundingBox, sceneStructs.h:84-87, 24 bytes */
    float lb[3];
    float ub[3];
} aipt_aabb;

/* ---- context ---------------------------------------------------------------------------------------------- */
/* stream: an existing hipStream_t to enqueue on (e.g. the caller's torch stream), or NULL to create one. */
int  aipt_create(int device, void* stream, aipt_ctx** out);
void aipt_destroy(aipt_ctx* ctx);
const char* aipt_last_error(const aipt_ctx* ctx);      /* ctx may be NULL: error of the last failed aipt_create */
int  aipt_abi_version(void);
int  aipt_sync(aipt_ctx* ctx);                          /* hipStreamSynchronize on the context stream */

/* device memory helpers for hosts that do not link HIP themselves */
int  aipt_malloc(aipt_ctx* ctx, size_t bytes, void** d_out);
int  aipt_free(aipt_ctx* ctx, void* d_ptr);
int  aipt_upload(aipt_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);      /* synchronous */
int  aipt_download(aipt_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);    /* synchronous */
int  aipt_memset(aipt_ctx* ctx, void* d_dst, int value, size_t bytes);              /* async */
/* HIP-event timer on the context stream (bench.py's roofline leg): start, ..., stop -> elapsed ms (stop syncs) */
int  aipt_timer_start(aipt_ctx* ctx);
int  aipt_timer_stop(aipt_ctx* ctx, float* ms_out);

/* ---- path trace (pathtrace.h:6-8) ------------------------------------------------------------------------- */
#define AIPT_TRACE_AA          1u   /* jitter primary rays (AA true, pathtrace.cu:25) */
#define AIPT_TRACE_COMPACT     2u   /* stream-compaction semantics (STREAM_COMPACTION true, pathtrace.cu:20): the RNG of
                                       bounce b is seeded with the path's rank among the live paths, as thrust::partition
                                       leaves it (pathtrace.cu:351,505).  Without it the seed index is the pixel index. */
#define AIPT_TRACE_RECORD_MAT0 4u   /* also record the first-hit material id per pixel (integer parity channel) */
#define AIPT_TRACE_BRUTE_FORCE 8u   /* mesh: test every face in index order like the reference (pathtrace.cu:258-269) instead
                                       of walking the BVH built at upload; same result, for parity checks and timing */
#define AIPT_TRACE_NO_BROAD_PHASE 16u /* primitives: run the exact box/sphere test of every primitive for every ray like the
                                       reference (pathtrace.cu:226-245) instead of only on the primitives whose padded
                                       world box the ray can touch; same result, for parity checks and timing */
#define AIPT_TRACE_SORT_MATERIAL 32u /* SORT_MATERIAL true (pathtrace.cu:21, 412-417, 508-510): after the partition the surviving paths
                                       are stably sorted by material id, exactly as the reference's thrust::sort_by_key call
                                       orders them (its keys are the hit records of the PRE-partition array slots; a miss has
                                       id 0), and the next bounce seeds each path's RNG with its slot after the sort.  Here the
                                       sort is a counting sort of the 4-byte live list (the path state does not move).  Needs
                                       AIPT_TRACE_COMPACT and at most 256 materials. */
#define AIPT_TRACE_CACHE_FIRST_BOUNCE 64u /* CACHE_BOUNCE true (pathtrace.cu:22, 466-476): iter == 1 saves the bounce-0 hit records,
                                       iter > 1 reuses them instead of intersecting the primary rays again.  Like the reference
                                       (assert, :435-436) only legal without AIPT_TRACE_AA and without AIPT_TRACE_MOTION_BLUR; the
                                       caller keeps camera and scene fixed between iter 1 and the iterations that reuse it. */
#define AIPT_TRACE_MOTION_BLUR 128u /* MOTION_BLUR true (pathtrace.cu:27, 318-331, 442-446): before every iteration with
                                       iter % 4 == 0 and iter < 3000 the primitives that have a velocity (Geom::vel, scene key VEL)
                                       move by vel * 0.10 and their matrices are rebuilt; the moved primitives persist in the
                                       context until the next aipt_scene_upload. */
#define AIPT_TRACE_NO_CULL     512u /* RAY_CULLING false (pathtrace.cu:23, 270-281): every ray is tested against the mesh, without the
                                       scene-AABB test in front (the test is not exact: rays that graze the box decide differently). */
#define AIPT_TRACE_DIELECTRIC  1024u /* DIELECTRIC true (interactions.h:6, 88-168, 179-192): materials scatter through Glass_BxDF /
                                       SpecularReflection_BxDF / SpecularRefraction_BxDF / Lambert_BxDF (Fresnel-weighted choice,
                                       0.001 ray offsets, un-normalised directions) instead of the Schlick branch. */
#define AIPT_TRACE_MESH_NORMAL_VIEW 2048u /* MESH_NORMAL_VIEW true (interactions.h:4, 222-255): the debug view -- a path's colour is
                                       multiplied by |surface normal| instead of the material colour (Schlick branch only). */
#define AIPT_TRACE_DEFAULT     (AIPT_TRACE_AA | AIPT_TRACE_COMPACT)

/* pathtraceInit (pathtrace.cu:96-129), scene part: copies and re-lays-out the scene on the device.
 * faces/mesh_box may be NULL when nfaces == 0. */
int aipt_scene_upload(aipt_ctx* ctx, const aipt_geom* geoms, int ngeoms, const aipt_material* materials, int nmaterials,
                      const aipt_face* faces, int nfaces, const aipt_aabb* mesh_box);
int aipt_scene_free(aipt_ctx* ctx);                      /* pathtraceFree (pathtrace.cu:131-145) */
/* The same upload in two steps, for multi-GPU hosts (no reference equivalent: the reference is single-GPU):
 * aipt_scene_pack (host only, no context, no GPU) validates the scene, builds the mesh BVH ONCE and returns one relocatable
 * blob -- geoms, materials, faces, BVH nodes, leaf triangle records -- that rank 0 broadcasts (RCCL) and every rank hands to
 * aipt_scene_upload_packed, which only copies.  aipt_scene_upload == pack + upload_packed + aipt_blob_free.
 * err (optional) receives the message when aipt_scene_pack fails. */
int  aipt_scene_pack(const aipt_geom* geoms, int ngeoms, const aipt_material* materials, int nmaterials,
                     const aipt_face* faces, int nfaces, const aipt_aabb* mesh_box, void** blob_out, size_t* bytes_out,
                     char* err, size_t errlen);
void aipt_blob_free(void* blob);
int  aipt_scene_upload_packed(aipt_ctx* ctx, const void* blob, size_t bytes);
/* pathtraceInit, frame-size part: path-state buffers for width x height pixels (allocated once, not per frame). */
int aipt_trace_configure(aipt_ctx* ctx, int width, int height);
/* pathtrace (pathtrace.cu:422-528), one iteration.  iter = 1 starts a new image (the interactive loop always does,
 * main.cpp:122-124,164); iter = 2, 3, ... with the same camera accumulate further samples into the context's image, planes
 * 0-2 become image / iter (multi-spp / ground-truth mode, main.cpp:147-151); planes 3-9 are written at iter == 1 only.  d_gbuf is float[10][gbuf_rows][gbuf_stride] with
 * gbuf_rows >= height and gbuf_stride >= width; the 10 planes are: 0-2 radiance/iter, 3-5 first-hit normal, 6 first-hit
 * distance, 7-9 first-bounce albedo, horizontally flipped exactly as copy_data / computeIntersections write them
 * (pathtrace.cu:81-94, 295-304, 379-387).  Every in-frame element is written each call; padding is left untouched. */
int aipt_trace(aipt_ctx* ctx, const aipt_camera* cam, int iter, int depth, uint32_t flags,
               float* d_gbuf, int gbuf_rows, int gbuf_stride);
/* Batched trace (no reference equivalent; the frame sequence is the data-parallel axis): nframes <= batch <= 24
 * iteration-1 frames with their own cameras are traced by ONE set of bounce launches -- a single 1280x720 frame leaves most of
 * the chip idle in the later bounces.  Frame f writes the G-buffer at d_gbuf + f * gbuf_frame_floats.  Every frame's
 * result is bit-identical to its own aipt_trace (the RNG index of a path is its rank among the live paths of ITS frame).  The
 * sort / cache / motion-blur toggles and iter > 1 are single-frame only.  aipt_trace_configure == batch 1. */
int aipt_trace_configure_batch(aipt_ctx* ctx, int width, int height, int batch);
int aipt_trace_batch(aipt_ctx* ctx, const aipt_camera* cams, int nframes, int iter, int depth, uint32_t flags,
                     float* d_gbuf, int gbuf_rows, int gbuf_stride, size_t gbuf_frame_floats);
int aipt_trace_live_counts_frame(aipt_ctx* ctx, int frame, int* h_n_live, int n);   /* of one frame of the last batch (aipt_frames: of its last trace call, both lanes) */
/* HIP-event timing of the bounce launches on the stream they are launched on (bench.py's roofline leg): up to max_calls
 * traces are recorded, every `every`-th one after _begin; _end synchronises and returns the summed ms of bounce b's launch in
 * sum_ms_per_bounce[b] (b < nbounces) and the number of recorded traces.  aipt_trace_kernel_name: the kernel instantiation
 * that ran bounce `bounce` of the last trace, as rocprofv3 names it. */
int aipt_trace_profile_begin(aipt_ctx* ctx, int max_calls, int every);
int aipt_trace_profile_end(aipt_ctx* ctx, double* sum_ms_per_bounce, int nbounces, int* calls);
/* per-call detail of the recorded traces, to be read BEFORE aipt_trace_profile_end (synchronises): call c held
 * nframes_per_call[c] frames and its bounce-b launch took ms_per_call_bounce[c * nbounces + b]; at most max_calls entries. */
int aipt_trace_profile_calls(aipt_ctx* ctx, int* nframes_per_call, double* ms_per_call_bounce, int nbounces, int max_calls, int* calls);
int aipt_trace_kernel_name(aipt_ctx* ctx, int bounce, char* kernel, size_t kernel_len);
/* BVH-walk statistics accumulated since the last reset, only in a library built with -DAIPT_TRACE_STATS (tools/trace_stats.py;
 * otherwise AIPT_E_STATE): out16 = {lane node visits, wave node-loop trips, lane triangle tests, wave leaf-loop trips, lane leaf
 * visits, max node visits of one ray, 0, 0, rays that walked with <= 4, 8, 16, 32, 64, 128, more node visits, 0}. */
int aipt_debug_trace_stats(aipt_ctx* ctx, unsigned long long* out16, int reset);
/* live-path counts of the last trace call: n_live[b] = paths entering bounce b, b = 0..depth, summed over the call's frames -- of
 * both lanes when aipt_frames traced the call's two halves side by side (synchronous). */
int aipt_trace_live_counts(aipt_ctx* ctx, int* h_n_live, int n);
/* first-hit material ids (-1 = miss) per path of the last AIPT_TRACE_RECORD_MAT0 trace (synchronous): n = pixels x frames of the
 * call; a batch is interleaved (path = pixel x frames + frame); after a two-lane aipt_frames call: the first lane's frames in that
 * layout, then the second lane's. */
int aipt_trace_first_hit_materials(aipt_ctx* ctx, int* h_mat, int n);

/* ---- denoiser (main.cpp:101-118; model = training/recurrent_autoencoder_model.py) --------------------------- */
#define AIPT_DN_BN_BATCH      1u    /* BatchNorm with statistics of the current frame: what the reference's shipped
                                       TorchScript computes (traced in train mode, convert_to_torchscript.py:26-30) */
#define AIPT_DN_BN_RUNNING    0u    /* BatchNorm with the stored running statistics (model.eval(), training/test.py:35) */
#define AIPT_DN_HIDDEN_CARRY  2u    /* carry the six recurrent hidden states from the previous call (forward(x, j>0)) */
#define AIPT_DN_HIDDEN_RESET  0u    /* zero hidden state (forward(x, j=0), recurrent_autoencoder_model.py:121-128) */

#define AIPT_DN_IMPL_MFMA     0     /* f32-input MFMA implicit-GEMM conv everywhere: bit-for-bit an fp32 FMA chain */
#define AIPT_DN_IMPL_VALU     1     /* plain per-thread direct conv (slow; on-GPU cross-check of the MFMA kernels) */
#define AIPT_DN_IMPL_MFMA_F16X3 2   /* DEFAULT: split-fp16 MFMA (hi/lo operands, 3 MFMAs per product, fp32-class accuracy) */
#define AIPT_DN_IMPL_MFMA_F16W 3    /* fp16 conv weights (BASELINE configs[4]): the weights are rounded to fp16, activations
                                       stay split hi/lo, fp32 accumulation (2 MFMAs per product, half the weight traffic);
                                       results = the reference model run with its conv weights rounded to fp16 */

/* HOST CONTRACT (gfx950 erratum): while a split-fp16 implementation (AIPT_DN_IMPL_MFMA_F16X3 / _F16W) runs, a
 * kernel of ANOTHER library that contains packed-fp32 VALU instructions (v_pk_*_f32; hipcc emits them for float2/float4
 * arithmetic and through its SLP vectoriser) and shares a CU with it can return wrong values in lanes 48..63 of a wave.  No
 * kernel of this library contains one (tests/test_no_packed_fp32_cpu.py), so the library is neither victim nor affected by
 * itself.  A host that runs such third-party kernels on the same GPU at the same time (torch ops on other streams, another
 * process) either orders them against aipt_* work (aipt_sync, events on the context's stream), keeps them on disjoint CUs
 * (hipExtStreamCreateWithCUMask), or opts out of the trigger with aipt_denoise_set_impl(ctx, AIPT_DN_IMPL_MFMA): fp32-input
 * MFMAs never cause it (at about half the denoiser throughput). */
/* blob: flat weight file, format in ai_path_tracer_denoiser_amd/arch.py (header + 28 x {W,b,gamma,beta,mean,var}).
 * Loading weights resets the recurrent hidden state (the next AIPT_DN_HIDDEN_CARRY frame starts from zeros). */
int aipt_denoise_load_weights(aipt_ctx* ctx, const void* blob, size_t bytes);
/* activations for frames of height x width (both multiples of 32: five 2x pools + skip concat). */
int aipt_denoise_configure(aipt_ctx* ctx, int height, int width);
int aipt_denoise_set_impl(aipt_ctx* ctx, int impl);
/* Operand range of the split-fp16 implementations (AIPT_DN_IMPL_MFMA_F16X3 / _F16W; AIPT_DN_IMPL_MFMA and _VALU are plain fp32).
 *  - network input (the G-buffer, in the caller's units): held to max(2^-22 |x|, 2^-32) absolute for |x| <= 65 504 x 2^4 =
 *    1 048 064 (1.0e6); larger values saturate there (on every size: both split-fp16 kernels clamp alike), and a frame whose
 *    EVERY plane is below ~1e-3 (unit normals never are) loses relative precision.
 *    The reference model is fp32 throughout (recurrent_autoencoder_model.py:8-142) and has neither bound.
 *  - BatchNorm sums are two-word fixed point: exact and order-independent while |sum x|, sum x^2 < 9.2e18 per channel and frame.
 *  - normalised activations y = LeakyReLU(BN(x)) are held as fp16 pairs: |y| < 4 094 on the levels of >= AIPT_DN_OPT_R_MINPIX
 *    pixels, < 65 504 below.  With AIPT_DN_BN_BATCH, |y| <= max|gamma| * sqrt(pixels) + max|beta| always holds, and a level whose
 *    bound from the LOADED weights exceeds its kernel's range runs on the next kernel down (finally the exact fp32 one): nothing
 *    for the caller to check.  With AIPT_DN_BN_RUNNING nothing bounds y: the caller keeps |y| < 4 094 or uses AIPT_DN_IMPL_MFMA
 *    (beyond the range a frame degrades to inf / NaN instead of fp32's large finite values).
 * Kernel selection of the split-fp16 implementations, by the pixel count of a level (defaults tuned on MI355X; every setting
 * computes the same function to <= 1e-3 and is covered by tests/test_gpu_denoise_kernels.py).  No environment variable changes
 * which arithmetic a caller gets. */
#define AIPT_DN_OPT_R_MINPIX     1   /* levels of >= value pixels: conv3x3_f16x3r, persistent register-staged (default 200000) */
#define AIPT_DN_OPT_F16_MINPIX   2   /* below R_MINPIX: conv3x3_f16x3 LDS-tiled, 8-row tiles from value pixels up, 4-row below (14000) --
                                        and 4-row on any level whose 4-row workgroups fit the chip at once (with KY_SPLIT) */
#define AIPT_DN_OPT_SMALL_MINPIX 3   /* levels below value pixels: the exact f32 MFMA kernel (default 0: none) */
#define AIPT_DN_OPT_FUSED_POOL   4   /* 1 (default): MaxPool2d(2) in the producing conv's epilogue; 0: a pool launch per encoder level */
#define AIPT_DN_OPT_KY_SPLIT      5   /* 1 (default): the 4-row tiles of conv3x3_f16x3 run three waves per row, one per tap row; 0: one */
int aipt_denoise_set_option(aipt_ctx* ctx, int option, long long value);
/* forward: d_in10 float[10][H][W] -> d_out3 float[3][H][W] */
int aipt_denoise(aipt_ctx* ctx, const float* d_in10, float* d_out3, uint32_t flags);
int aipt_denoise_reset_hidden(aipt_ctx* ctx);
/* checkpoint/resume of the recurrent state: level 0..4 = encoder1..5, 5 = bottleneck; d_buf float[C][H>>l][W>>l]. */
int aipt_denoise_get_hidden(aipt_ctx* ctx, int level, float* d_dst);
int aipt_denoise_set_hidden(aipt_ctx* ctx, int level, const float* d_src);

/* per-conv-layer HIP-event timing on the context stream (bench.py's roofline leg).  layer_mask: bit l = time layer l
 * (0..27, network order); up to max_calls forward passes are recorded; _end synchronises and returns the summed ms. */
int aipt_denoise_profile_begin(aipt_ctx* ctx, uint32_t layer_mask, int max_calls);
int aipt_denoise_profile_end(aipt_ctx* ctx, double* sum_ms28, int* calls);
/* time only every `every`-th forward pass after _begin (default 1): each event pair costs the stream ~2 us */
int aipt_denoise_profile_stride(aipt_ctx* ctx, int every);
/* kernel instantiation that ran `layer` in the last forward, its shape and algorithmic FLOPs (2*9*cin*cout*h*w) */
int aipt_denoise_layer_info(aipt_ctx* ctx, int layer, char* kernel, size_t kernel_len, int* cin, int* cout,
                            int* height, int* width, double* flops);

/* ---- one frame: trace -> device G-buffer -> denoise (runCuda body, main.cpp:143-163) ------------------------ */
/* Requires scene upload, aipt_frame_configure, weights.  Frames whose size is not a multiple of 32 are zero-padded
 * at the bottom/right to the next multiple (the reference model cannot run them at all); d_out3 is
 * float[3][height][width] (cropped). */
int aipt_frame_configure(aipt_ctx* ctx, int width, int height);
int aipt_frame(aipt_ctx* ctx, const aipt_camera* cam, int iter, int depth, uint32_t trace_flags, uint32_t dn_flags,
               float* d_out3);
/* Throughput pipelining for frame-by-frame hosts (no reference equivalent: runCuda traces and denoises strictly in turn,
 * main.cpp:143-163).  Starts the path trace of the NEXT frame into the context's back G-buffer on a stream restricted to
 * 3/8 of the CUs; the following aipt_frame with an identical (cam, iter, depth, trace_flags) consumes it instead of tracing
 * and runs its denoise on a stream restricted to the OTHER CUs, so that the next prefetch's trace runs beside it.  A
 * different request drops the prefetch.  Results are identical to frames without it (one frame of latency, +27 % frames/s
 * on the mesh configuration: 782 against 616; on scenes whose trace is cheap the reduced denoiser costs more than the overlap
 * gains: do not prefetch there).  Round 5: 96 CUs trace : 160 CUs denoise (rounds 2-4: halves); the persistent conv kernel is
 * launched with one workgroup per CU of ITS share (same bits at any workgroup count).  The two streams use DISJOINT CUs as a scheduling choice: a single frame's trace does not fill the chip, and
 * sharing all CUs measured slower.  (In round 2 the split was also a fence: a bounce kernel sharing a CU with the split-fp16
 * conv kernel returned wrong values in a few lanes -- packed-fp32 VALU instructions beside gapped fp16 MFMAs, a gfx950 erratum;
 * the library is built without packed fp32 since round 3 and no kernel of it can be the victim)
 * AIPT_PREFETCH_TRACE_CUS overrides the split.  Only iter == 1 frames can be prefetched
 * (planes 3-9 of later iterations live in the buffer iteration 1 wrote): other values return AIPT_E_INVALID.  aipt_sync waits
 * for every stream of the context; work queued on the context's stream after aipt_frame sees its result. */
int aipt_frame_prefetch(aipt_ctx* ctx, const aipt_camera* cam, int iter, int depth, uint32_t trace_flags);
/* Frame batches: aipt_frames_configure(batch <= 32) after aipt_frame_configure; aipt_frames traces nframes consecutive frames
 * with one set of launches per up to 24 frames (32 frames: 16 + 16) (aipt_trace_batch) and then denoises them in order -- frame 0 with dn_flags_first, the
 * others with dn_flags_rest (e.g. carry the hidden state inside the batch) -- into d_out3[0..nframes).  The denoiser passes of
 * consecutive frames run on two streams, frame n+1 entering an encoder level when frame n has left it (its hidden state of
 * that level is written); everything is joined on the context's stream before the call returns.  Same results, bit for bit,
 * as nframes calls of aipt_frame.  aipt_frame_last_times then reports per-frame averages over the batch. */
int aipt_frames_configure(aipt_ctx* ctx, int batch);
int aipt_frames(aipt_ctx* ctx, const aipt_camera* cams, int nframes, int iter, int depth, uint32_t trace_flags,
                uint32_t dn_flags_first, uint32_t dn_flags_rest, float* const* d_out3);
int aipt_frames_gbuffer(aipt_ctx* ctx, int frame, float** d_gbuf, int* rows, int* stride);
/* for batches: queues the traces of the NEXT batch into the back set of G-buffers BEHIND the denoiser passes of the current
 * one, on the context's stream (a batch's traces fill the chip: nothing to gain from a CU split, and they must not share CUs
 * with conv kernels); the following aipt_frames with identical (cams, iter, depth, trace_flags) consumes it. */
int aipt_frames_prefetch(aipt_ctx* ctx, const aipt_camera* cams, int nframes, int iter, int depth, uint32_t trace_flags);
/* the context-owned padded G-buffer float[10][Hp][Wp] of the last aipt_frame (device pointer) and its padded size */
int aipt_gbuffer(aipt_ctx* ctx, float** d_gbuf, int* rows, int* stride);
/* per-stage GPU time of the last aipt_frame with timing enabled (ms; synchronous) */
int aipt_frame_set_timing(aipt_ctx* ctx, int enabled);
int aipt_frame_last_times(aipt_ctx* ctx, float* trace_ms, float* denoise_ms);

/* ---- multi-GPU hosts: the one-time broadcast (no reference equivalent: single GPU) ------------------------------ */
/* The path shards by frames (one context per GPU, contiguous frame chunks, no per-frame exchange); the only collective is
 * rank 0's packed scene (aipt_scene_pack: geometry + the BVH built once) and weight blob going to every rank.
 * aipt_comm_create over n contexts: RCCL (ncclCommInitAll, grouped ncclBroadcast over xGMI; librccl is dlopen'ed) when every
 * context has its own GPU, an in-process copy shim when contexts share a GPU or force_shim != 0 (`aiptd --gpus 1 --ranks 8`).
 * aipt_comm_broadcast: d_bufs[r] = device buffer of `bytes` bytes in context r; after the call all hold root's bytes
 * (synchronous).  Single host thread. */
typedef struct aipt_comm aipt_comm;
int  aipt_comm_create(aipt_ctx* const* ctxs, int n, int force_shim, aipt_comm** out);
int  aipt_comm_is_rccl(const aipt_comm* comm);
int  aipt_comm_broadcast(aipt_comm* comm, void* const* d_bufs, size_t bytes, int root);
void aipt_comm_destroy(aipt_comm* comm);
int  aipt_device_count(void);

/* ---- host-side scene front end (scene.cpp:11-320, utilities.cpp:45-52, main.cpp:66-78,122-140) -------------- */
typedef struct aipt_scene aipt_scene;
/* Parses the reference's scene grammar (MATERIAL / OBJECT / CAMERA / MESH blocks), builds the three matrices of every
 * primitive, loads + transforms the OBJ mesh and its bounding box, and sets the first-frame orbit camera.
 * err (optional) receives the message on failure. */
int  aipt_scene_load(const char* path, aipt_scene** out, char* err, size_t errlen);
/* The same with the loader's compile-time switch as a flag.  AIPT_SCENE_RECOMPUTE_NORMALS = RECOMPUTE_NORMALS true (scene.cpp:9,
 * 198-204, 310-311): the OBJ's vertex normals are replaced by the face normal normalize(cross(v2 - v0, v1 - v0)) of the
 * transformed triangle -- in n[0] and n[1]; n[2] stays the zero vector, as the reference's assignment leaves it. */
#define AIPT_SCENE_RECOMPUTE_NORMALS 1u
int  aipt_scene_load_ex(const char* path, unsigned flags, aipt_scene** out, char* err, size_t errlen);
void aipt_scene_release(aipt_scene* scene);
int  aipt_scene_set_resolution(aipt_scene* scene, int width, int height);   /* override RES; recomputes fov/pixelLength */
int  aipt_scene_info(const aipt_scene* scene, int* ngeoms, int* nmaterials, int* nfaces, int* iterations, int* depth);
const aipt_geom*     aipt_scene_geoms(const aipt_scene* scene);
const aipt_material* aipt_scene_materials(const aipt_scene* scene);
const aipt_face*     aipt_scene_faces(const aipt_scene* scene);
const aipt_aabb*     aipt_scene_mesh_box(const aipt_scene* scene);
int  aipt_scene_camera(const aipt_scene* scene, aipt_camera* cam_out);        /* first-frame camera */
int  aipt_scene_orbit_params(const aipt_scene* scene, float* zoom, float* phi, float* theta);   /* as main() derives them */
int  aipt_scene_upload_host(aipt_ctx* ctx, const aipt_scene* scene);          /* aipt_scene_upload of the parsed arrays */
/* transform / inverseTransform / invTranspose of one primitive from its translation, rotation (degrees), scale */
void aipt_geom_build(aipt_geom* geom);
/* orbit camera: parameters from a loaded camera (main.cpp:66-78) and the per-frame rebuild (main.cpp:122-140) */
void aipt_camera_orbit_params(const aipt_camera* cam, float* zoom, float* phi, float* theta);
void aipt_camera_orbit(aipt_camera* cam, float zoom, float phi, float theta);

#ifdef __cplusplus
}
#endif
#endif /* AIPTD_H */
