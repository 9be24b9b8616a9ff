// pathtrace_aiptd.cpp -- what a maintainer of the reference compiles INSTEAD of Inference/src/pathtrace.cu (and instead of
// the libtorch call in main.cpp:101-118) to run the hot path on libaiptd.so.  This is the code INTEGRATION.md shows; it is
// compiled and linked by tests/test_integration_shim.py against tests/integration/mock/ (stand-ins for pathtrace.h /
// sceneStructs.h) and, on a GPU box, run end to end.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pathtrace.h"        // void pathtraceInit(Scene*), pathtraceFree(), pathtrace(uchar4*, int, int)   (pathtrace.h:6-8)
#include "aiptd.h"            // this repo: include/aiptd.h, link -laiptd

static aipt_ctx* g_ctx = nullptr;
static Scene*    g_scene = nullptr;
static float*    d_out3 = nullptr;              // denoised RGB [3][H][W] on the device

static_assert(sizeof(Geom) == sizeof(aipt_geom) && sizeof(Material) == sizeof(aipt_material) &&
              sizeof(Face) == sizeof(aipt_face) && sizeof(Camera) == sizeof(aipt_camera) &&
              sizeof(MeshBoundingBox) == sizeof(aipt_aabb), "POD layouts (sceneStructs.h)");

void pathtraceInit(Scene* scene) {               // pathtrace.cu:96-129
    g_scene = scene;
    if (!g_ctx && aipt_create(0, nullptr, &g_ctx) != AIPT_OK) { fprintf(stderr, "%s\n", aipt_last_error(nullptr)); exit(1); }
    const Camera& cam = scene->state.camera;
    int rc = aipt_scene_upload(g_ctx, (const aipt_geom*)scene->geoms.data(), (int)scene->geoms.size(),
                               (const aipt_material*)scene->materials.data(), (int)scene->materials.size(),
                               (const aipt_face*)scene->faces.data(), (int)scene->faces.size(),
                               (const aipt_aabb*)&scene->mesh_box);
    if (!rc) rc = aipt_frame_configure(g_ctx, cam.resolution.x, cam.resolution.y);   // no-op when the size is unchanged
    if (!rc && !d_out3) rc = aipt_malloc(g_ctx, sizeof(float) * 3 * cam.resolution.x * cam.resolution.y, (void**)&d_out3);
    if (rc) { fprintf(stderr, "pathtraceInit: %s\n", aipt_last_error(g_ctx)); exit(1); }
}
void pathtraceFree() { if (g_ctx) aipt_scene_free(g_ctx); }            // pathtrace.cu:131-145

void pathtrace(uchar4* /*pbo*/, int /*frame*/, int iter) {             // pathtrace.cu:422-528
    const Camera& cam = g_scene->state.camera;
    // trace + denoise with the G-buffer kept on the device; TorchScript semantics = batch-stat BN, zero hidden
    const int rc = aipt_frame(g_ctx, (const aipt_camera*)&cam, iter, g_scene->state.traceDepth, AIPT_TRACE_DEFAULT,
                              AIPT_DN_BN_BATCH | AIPT_DN_HIDDEN_RESET, d_out3);
    if (rc) { fprintf(stderr, "pathtrace: %s\n", aipt_last_error(g_ctx)); exit(1); }
    // the reference hands the 10-channel tensor to main.cpp through host memory (pathtrace.cu:525); keep that contract
    float* d_gbuf; int rows, stride;
    if (aipt_gbuffer(g_ctx, &d_gbuf, &rows, &stride)) { fprintf(stderr, "pathtrace: %s\n", aipt_last_error(g_ctx)); exit(1); }
    const int W = cam.resolution.x, H = cam.resolution.y;
    int drc;
    if (rows == H && stride == W) {
        drc = aipt_download(g_ctx, g_scene->state.host_tensor, d_gbuf, sizeof(float) * 10 * rows * stride);
    } else {
        // frame sizes that are not multiples of 32: the device G-buffer is zero-padded at the bottom / right (the reference
        // model cannot run such frames at all); host_tensor keeps the reference's float[10][H][W]
        static std::vector<float> padded;
        padded.resize((size_t)10 * rows * stride);
        drc = aipt_download(g_ctx, padded.data(), d_gbuf, sizeof(float) * padded.size());
        for (int c = 0; c < 10 && !drc; c++)
            for (int y = 0; y < H; y++)
                memcpy(g_scene->state.host_tensor + ((size_t)c * H + y) * W, padded.data() + ((size_t)c * rows + y) * stride, sizeof(float) * W);
    }
    if (drc) { fprintf(stderr, "pathtrace: %s\n", aipt_last_error(g_ctx)); exit(1); }
}

// main.cpp:101-118 network_prediction_faster_version(float*) becomes a download of the denoised frame, [3][H][W]
void aiptd_denoised_frame(float* h_rgb) {
    const Camera& cam = g_scene->state.camera;
    aipt_download(g_ctx, h_rgb, d_out3, sizeof(float) * 3 * cam.resolution.x * cam.resolution.y);
}

// main.cpp:107 torch::jit::load(MODEL_PATH) becomes one call at start-up with the blob tools/export_weights.py wrote
int aiptd_load_weights(const void* blob, size_t bytes) {
    if (!g_ctx && aipt_create(0, nullptr, &g_ctx) != AIPT_OK) { fprintf(stderr, "%s\n", aipt_last_error(nullptr)); return -1; }
    return aipt_denoise_load_weights(g_ctx, blob, bytes);
}
