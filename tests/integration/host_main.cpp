// A miniature of the reference's host (main.cpp: scene -> pathtraceInit -> pathtrace per frame -> host_tensor + denoised image)
// driving tests/integration/pathtrace_shim.cpp through the reference's three prototypes.  The Scene is filled by the library's
// own scene-file front end; usage: host_main SCENE.txt WEIGHTS.aiptw OUT.f32 (writes 10*W*H G-buffer floats + 3*W*H denoised).
#include <cstdio>
#include <cstring>
#include <vector>

#include "pathtrace.h"
#include "aiptd.h"

void aiptd_denoised_frame(float* h_rgb);
int aiptd_load_weights(const void* blob, size_t bytes);

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: host_main SCENE.txt WEIGHTS OUT\n"); return 2; }
    aipt_scene* s = nullptr;
    char err[256];
    if (aipt_scene_load(argv[1], &s, err, sizeof(err))) { fprintf(stderr, "%s\n", err); return 1; }
    int ng, nm, nf, it, depth;
    aipt_scene_info(s, &ng, &nm, &nf, &it, &depth);
    Scene scene;
    scene.geoms.resize(ng); scene.materials.resize(nm); scene.faces.resize(nf);
    if (ng) memcpy(scene.geoms.data(), aipt_scene_geoms(s), sizeof(Geom) * ng);
    memcpy(scene.materials.data(), aipt_scene_materials(s), sizeof(Material) * nm);
    if (nf) memcpy(scene.faces.data(), aipt_scene_faces(s), sizeof(Face) * nf);
    memcpy(&scene.mesh_box, aipt_scene_mesh_box(s), sizeof(MeshBoundingBox));
    aipt_scene_camera(s, (aipt_camera*)&scene.state.camera);
    scene.state.traceDepth = depth;
    const int W = scene.state.camera.resolution.x, H = scene.state.camera.resolution.y;
    std::vector<float> tensor((size_t)10 * W * H), rgb((size_t)3 * W * H);
    scene.state.host_tensor = tensor.data();
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<char> blob(n);
    if (fread(blob.data(), 1, n, f) != (size_t)n) return 1;
    fclose(f);
    if (aiptd_load_weights(blob.data(), blob.size())) return 1;
    pathtraceInit(&scene);
    pathtrace(nullptr, 0, 1);
    aiptd_denoised_frame(rgb.data());
    pathtraceFree();
    f = fopen(argv[3], "wb");
    fwrite(tensor.data(), 4, tensor.size(), f);
    fwrite(rgb.data(), 4, rgb.size(), f);
    fclose(f);
    printf("%d %d\n", W, H);
    return 0;
}
