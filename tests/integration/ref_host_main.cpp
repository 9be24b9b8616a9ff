// The reference's OWN host front end driving tests/integration/pathtrace_shim.cpp -- the drop-in, in the reference's tree.
//
// Compiled only where /root/reference exists (oracle/Makefile target `ref` -> oracle/_ref/shim_ref_host, git-ignored) from:
//   * the reference's unmodified Inference/src/scene.cpp + utilities.cpp + vendored tinyobj / GLM (its Scene class parses the
//     scene file: nothing of this repository's parser is involved),
//   * the reference's unmodified Inference/src/pathtrace.h / scene.h / sceneStructs.h (the three prototypes and the structs the
//     shim is written against; <cuda_runtime.h> = the genuine NVIDIA header inside triton's NVIDIA backend),
//   * tests/integration/pathtrace_shim.cpp (what replaces pathtrace.cu + the libtorch call) and libaiptd.so.
// This file restates the few lines of main.cpp that sit between them (main.cpp itself includes GLFW / libtorch / OpenCV /
// <Windows.h>): :59-78 load + orbit parameters, :122-140 the camchanged block, :143-163 init -> pathtrace -> denoise.
// Usage: shim_ref_host SCENE.txt WEIGHTS.aiptw OUT.f32 [W H]   (writes 10*W*H G-buffer floats + 3*W*H denoised floats)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define TINYOBJLOADER_IMPLEMENTATION        // main.cpp:4
#include "tiny_obj_loader.h"

#include "pathtrace.h"                      // the reference's header

void aiptd_denoised_frame(float* h_rgb);
int aiptd_load_weights(const void* blob, size_t bytes);

int main(int argc, char** argv) {
    if (argc != 4 && argc != 6) { fprintf(stderr, "usage: shim_ref_host SCENE.txt WEIGHTS OUT [W H]\n"); return 2; }
    std::streambuf* keep = std::cout.rdbuf(std::cerr.rdbuf());          // the reference's parser chats on stdout
    Scene* scene = new Scene(argv[1]);                                  // main.cpp:59
    std::cout.rdbuf(keep);
    RenderState* renderState = &scene->state;
    Camera& cam = renderState->camera;
    if (argc == 6) {
        // a test-sized frame: what editing the RES line of the scene file does (scene.cpp:113-115, 143-157)
        const float fovy = cam.fov.y;
        cam.resolution.x = atoi(argv[4]); cam.resolution.y = atoi(argv[5]);
        float yscaled = tan(fovy * (PI / 180));
        float xscaled = (yscaled * cam.resolution.x) / cam.resolution.y;
        float fovx = (atan(xscaled) * 180) / PI;
        cam.fov = glm::vec2(fovx, fovy);
        cam.pixelLength = glm::vec2(2 * xscaled / (float)cam.resolution.x, 2 * yscaled / (float)cam.resolution.y);
        delete[] renderState->host_tensor;
        renderState->host_tensor = new float[(size_t)cam.resolution.x * cam.resolution.y * 10];
    }
    // main.cpp:66-78
    glm::vec3 view = cam.view;
    glm::vec3 viewXZ = glm::vec3(view.x, 0.0f, view.z);
    glm::vec3 viewZY = glm::vec3(0.0f, view.y, view.z);
    float phi = glm::acos(glm::dot(glm::normalize(viewXZ), glm::vec3(0, 0, -1)));
    float theta = glm::acos(glm::dot(glm::normalize(viewZY), glm::vec3(0, 1, 0)));
    float zoom = glm::length(cam.position - cam.lookAt);
    // main.cpp:122-140 (camchanged is true on the first frame)
    glm::vec3 cameraPosition;
    cameraPosition.x = zoom * sin(phi) * sin(theta);
    cameraPosition.y = zoom * cos(theta);
    cameraPosition.z = zoom * cos(phi) * sin(theta);
    cam.view = -glm::normalize(cameraPosition);
    glm::vec3 v = cam.view;
    glm::vec3 u = glm::vec3(0, 1, 0);
    glm::vec3 r = glm::cross(v, u);
    cam.up = glm::cross(r, v);
    cam.right = r;
    cameraPosition += cam.lookAt;
    cam.position = cameraPosition;

    FILE* f = fopen(argv[2], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blob(n);
    if (fread(blob.data(), 1, n, f) != (size_t)n) return 1;
    fclose(f);
    if (aiptd_load_weights(blob.data(), blob.size())) return 1;        // main.cpp:107 torch::jit::load
    const int W = cam.resolution.x, H = cam.resolution.y;
    std::vector<float> rgb((size_t)3 * W * H);
    // main.cpp:143-163
    pathtraceFree();
    pathtraceInit(scene);
    pathtrace(NULL, 0, 1);
    aiptd_denoised_frame(rgb.data());                                   // network_prediction_faster_version(host_tensor)
    pathtraceFree();
    f = fopen(argv[3], "wb");
    fwrite(renderState->host_tensor, 4, (size_t)10 * W * H, f);
    fwrite(rgb.data(), 4, rgb.size(), f);
    fclose(f);
    printf("%d %d\n", W, H);
    return 0;
}
