// ref_scene_dump.cpp -- dumps what the REFERENCE'S OWN scene front end makes of a scene file (test infrastructure).
//
// Compiles, where they lie under /root/reference/Inference, the reference's unmodified src/scene.cpp (Scene::Scene,
// loadGeom, loadCamera, loadMaterial, loadObj: :11-320), src/utilities.cpp (tokenizeString, safeGetline,
// buildTransformationMatrix) and the vendored tiny_obj_loader.h / GLM, with plain g++ against the image's genuine NVIDIA
// <cuda_runtime.h> (see ref_isect_kats.cpp).  Recipe: oracle/Makefile target `ref` -> oracle/_ref/scene_dump (git-ignored).
// Built at -O0: Scene::loadObj / loadGeom / loadCamera fall off the end of non-void functions on some paths
// (scene.cpp:320), which g++ -O1 and above turns into a crash.  Nothing of the reference is copied: this driver constructs
// the reference's Scene, then restates the few lines of main() that derive the orbit camera (they live in main.cpp beside
// GLFW / libtorch / OpenCV / <Windows.h> includes and cannot be compiled from there) and writes raw bytes to stdout.
// tests/golden/gen_trace_kats.py turns the dumps into tests/golden/scene_ref_dumps.npz, which pins the oracle's parser
// (oracle/__init__.py OracleScene) and the product's aipt_scene_load / aipt_camera_orbit (tests/test_scene_frontend.py).
//
// main.cpp lines restated here (and only here):
//   :59-78    scene = new Scene(file); view/up/right; viewXZ, viewZY; phi, theta = acos(dot(normalize(..), axis)); zoom
//   :122-140  runCuda()'s camchanged block: cameraPosition from (zoom, phi, theta); cam.view/up/right; cam.position
//
// Output (little-endian, packed):
//   int32  ngeoms, nmaterials, nfaces, iterations, traceDepth
//   Geom[ngeoms] (248 B) | Material[nmaterials] (44 B) | Face[nfaces] (76 B) | MeshBoundingBox (24 B)
//   Camera (84 B, as loaded) | float zoom, phi, theta
//   int32 norbit | norbit x { float dphi, dtheta; Camera (84 B) after the camchanged block at (zoom, phi + dphi, theta + dtheta) }
// Usage: scene_dump <scene.txt> [dphi dtheta]...     (run with the scene's directory as cwd: MESH PATHs are relative)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define TINYOBJLOADER_IMPLEMENTATION        // main.cpp:4 does this for the reference's build
#include "tiny_obj_loader.h"

#include "scene.h"                          // the reference's header, unmodified

static_assert(sizeof(Geom) == 248 && sizeof(Material) == 44 && sizeof(Face) == 76 && sizeof(Camera) == 84 &&
              sizeof(MeshBoundingBox) == 24, "sceneStructs.h layouts");

template <class T> static void put(const T& v) { fwrite(&v, sizeof(T), 1, stdout); }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    // the reference's constructor chats on stdout: keep the dump clean by parking stdout on stderr meanwhile
    fflush(stdout);
    std::streambuf* keep = std::cout.rdbuf(std::cerr.rdbuf());
    Scene* scene = new Scene(argv[1]);      // ~Scene is declared, never defined: never deleted (as main.cpp)
    std::cout.rdbuf(keep);

    RenderState* renderState = &scene->state;
    Camera& cam = renderState->camera;
    // main.cpp:66-78
    glm::vec3 view = cam.view;
    glm::vec3 up = cam.up;
    glm::vec3 right = glm::cross(view, up);
    up = glm::cross(right, view);
    glm::vec3 cameraPosition = cam.position;
    glm::vec3 viewXZ = glm::vec3(view.x, 0.0f, view.z);
    glm::vec3 viewZY = glm::vec3(0.0f, view.y, view.z);
    float phi = glm::acos(glm::dot(glm::normalize(viewXZ), glm::vec3(0, 0, -1)));
    float theta = glm::acos(glm::dot(glm::normalize(viewZY), glm::vec3(0, 1, 0)));
    glm::vec3 ogLookAt = cam.lookAt;
    float zoom = glm::length(cam.position - ogLookAt);

    put<int32_t>((int32_t)scene->geoms.size());
    put<int32_t>((int32_t)scene->materials.size());
    put<int32_t>((int32_t)scene->faces.size());
    put<int32_t>((int32_t)renderState->iterations);
    put<int32_t>((int32_t)renderState->traceDepth);
    for (const Geom& g : scene->geoms) put(g);
    for (const Material& m : scene->materials) put(m);
    for (const Face& f : scene->faces) put(f);
    put(scene->mesh_box);
    put(cam);
    put(zoom); put(phi); put(theta);

    const int norbit = (argc - 2) / 2;
    put<int32_t>(norbit);
    const Camera loaded = cam;
    for (int k = 0; k < norbit; ++k) {
        const float dphi = (float)atof(argv[2 + 2 * k]), dtheta = (float)atof(argv[3 + 2 * k]);
        cam = loaded;
        const float p = phi + dphi, t = theta + dtheta;
        // main.cpp:122-140
        cameraPosition.x = zoom * sin(p) * sin(t);
        cameraPosition.y = zoom * cos(t);
        cameraPosition.z = zoom * cos(p) * sin(t);
        cam.view = -glm::normalize(cameraPosition);
        glm::vec3 v = cam.view;
        glm::vec3 u = glm::vec3(0, 1, 0);
        glm::vec3 r = glm::cross(v, u);
        cam.up = glm::cross(r, v);
        cam.right = r;
        cameraPosition += cam.lookAt;
        cam.position = cameraPosition;
        put(dphi); put(dtheta);
        put(cam);
    }
    fflush(stdout);
    return 0;
}
