// Stand-in for the reference's Inference/src/sceneStructs.h, for COMPILING tests/integration/pathtrace_shim.cpp outside the
// reference tree.  Not a copy: only the members the shim touches, with plain float aggregates where the reference uses
// glm::vec3 / glm::mat4 / glm::ivec2 (same sizes and offsets: the static_asserts in the shim check every struct against
// include/aiptd.h, whose layouts tests/test_abi_cpu.py pins to the reference's [probed] sizes 248 / 76 / 44 / 84 / 24).
#pragma once
#include <string>
#include <vector>

struct vec3 { float x, y, z; };
struct ivec2 { int x, y; };
struct mat4 { float m[16]; };

struct Geom { int type; int materialid; vec3 translation, rotation, scale; mat4 transform, inverseTransform, invTranspose; vec3 vel; };
struct Face { vec3 v[3]; vec3 n[3]; int materialid; };
struct Material { vec3 color; struct { float exponent; vec3 color; } specular; float hasReflective, hasRefractive, indexOfRefraction, emittance; };
struct Camera { ivec2 resolution; vec3 position, lookAt, view, up, right; float fov[2]; float pixelLength[2]; };
struct MeshBoundingBox { vec3 lb, ub; };
// the reference's full member list, in its order (sceneStructs.h:69-75): the shim is compiled against the layout it will meet
struct RenderState { Camera camera; unsigned int iterations; int traceDepth; std::string imageName; float* host_tensor; };

struct Scene {                      // scene.h:13-44: the members pathtrace.cu reads
    std::vector<Geom> geoms;
    std::vector<Material> materials;
    std::vector<Face> faces;
    MeshBoundingBox mesh_box;
    RenderState state;
};
