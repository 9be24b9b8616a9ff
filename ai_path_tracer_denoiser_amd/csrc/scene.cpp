// scene.cpp -- host-side scene front end of libaiptd.so: the reference's scene-file grammar, transform
// and camera math, OBJ mesh loading and the mesh bounding box.
//
// Replaces Scene::Scene / loadMaterial / loadGeom / loadCamera / loadObj (reference Inference/src/scene.cpp:11-320),
// utilityCore::buildTransformationMatrix (utilities.cpp:45-52) and the orbit-camera rebuild of runCuda()
// (main.cpp:66-78, 122-140).  Matrix products follow GLM 0.9.6.3's statement order (gtc/matrix_transform.inl:40-134,
// detail/type_mat4x4.inl:37-92 and :685-703, gtc/matrix_inverse.inl:95-147) so the uploaded matrices are the floats the
// reference would upload.  Compiled with -ffp-contract=off.
//
// Grammar quirks kept on purpose: a MATERIAL block is exactly 7 lines and the CAMERA header exactly 5 (scene.cpp:171,
// :109); unknown keys are ignored; FOVY is used as the half-angle (scene.cpp:143); camera.right is left un-normalised by
// the orbit rebuild (main.cpp:133-135); the mesh box's upper bound starts at FLT_MIN, the smallest positive float, not
// the lowest (scene.cpp:216-218); vertex normals are normalised but NOT rotated (scene.cpp:304-307).
#include "internal.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

struct aipt_scene {
    std::vector<aipt_geom> geoms;
    std::vector<aipt_material> materials;
    std::vector<aipt_face> faces;
    aipt_aabb box{};
    aipt_camera cam{};
    float fovy = 45.0f;
    int iterations = 1, depth = 8;
    float zoom = 0, phi = 0, theta = 0;   // orbit parameters derived at load (main.cpp:66-78)
    std::string image_name;
    std::string dir;        // directory of the scene file (mesh paths are resolved against it, then as given)
};

namespace {

constexpr float kPi = 3.1415926535897932384626422832795028841971f;

struct f3 { float x, y, z; };
inline f3 F3(float x, float y, float z) { return f3{x, y, z}; }
inline float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline f3 scale3(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
inline f3 sub3(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline f3 cross3(f3 x, f3 y) { return F3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
inline f3 normalize3(f3 a) { return scale3(a, 1.0f / std::sqrt(dot3(a, a))); }

struct M4 { float m[16]; };   // column-major, m[c*4+r]

M4 identity() { M4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }

M4 mul(const M4& a, const M4& b) {
    M4 r;
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++)
            r.m[c * 4 + row] = ((a.m[row] * b.m[c * 4] + a.m[4 + row] * b.m[c * 4 + 1]) + a.m[8 + row] * b.m[c * 4 + 2])
                               + a.m[12 + row] * b.m[c * 4 + 3];
    return r;
}

M4 translate(const M4& m, const float* v) {
    M4 r = m;
    for (int row = 0; row < 4; row++)
        r.m[12 + row] = ((m.m[row] * v[0] + m.m[4 + row] * v[1]) + m.m[8 + row] * v[2]) + m.m[12 + row];
    return r;
}

M4 rotate(const M4& m, float angle, f3 v) {
    const float c = std::cos(angle), s = std::sin(angle);
    const f3 axis = normalize3(v);
    const f3 temp = scale3(axis, 1.0f - c);
    float R[3][3];
    R[0][0] = c + temp.x * axis.x;
    R[0][1] = 0 + temp.x * axis.y + s * axis.z;
    R[0][2] = 0 + temp.x * axis.z - s * axis.y;
    R[1][0] = 0 + temp.y * axis.x - s * axis.z;
    R[1][1] = c + temp.y * axis.y;
    R[1][2] = 0 + temp.y * axis.z + s * axis.x;
    R[2][0] = 0 + temp.z * axis.x + s * axis.y;
    R[2][1] = 0 + temp.z * axis.y - s * axis.x;
    R[2][2] = c + temp.z * axis.z;
    M4 r;
    for (int j = 0; j < 3; j++)
        for (int row = 0; row < 4; row++)
            r.m[j * 4 + row] = (m.m[row] * R[j][0] + m.m[4 + row] * R[j][1]) + m.m[8 + row] * R[j][2];
    for (int row = 0; row < 4; row++) r.m[12 + row] = m.m[12 + row];
    return r;
}

M4 scale(const M4& m, const float* v) {
    M4 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 4; row++) r.m[c * 4 + row] = m.m[c * 4 + row] * v[c];
    for (int row = 0; row < 4; row++) r.m[12 + row] = m.m[12 + row];
    return r;
}

#define E(c, r) a.m[(c) * 4 + (r)]
M4 inverse(const M4& a) {
    const float c00 = E(2,2) * E(3,3) - E(3,2) * E(2,3), c02 = E(1,2) * E(3,3) - E(3,2) * E(1,3), c03 = E(1,2) * E(2,3) - E(2,2) * E(1,3);
    const float c04 = E(2,1) * E(3,3) - E(3,1) * E(2,3), c06 = E(1,1) * E(3,3) - E(3,1) * E(1,3), c07 = E(1,1) * E(2,3) - E(2,1) * E(1,3);
    const float c08 = E(2,1) * E(3,2) - E(3,1) * E(2,2), c10 = E(1,1) * E(3,2) - E(3,1) * E(1,2), c11 = E(1,1) * E(2,2) - E(2,1) * E(1,2);
    const float c12 = E(2,0) * E(3,3) - E(3,0) * E(2,3), c14 = E(1,0) * E(3,3) - E(3,0) * E(1,3), c15 = E(1,0) * E(2,3) - E(2,0) * E(1,3);
    const float c16 = E(2,0) * E(3,2) - E(3,0) * E(2,2), c18 = E(1,0) * E(3,2) - E(3,0) * E(1,2), c19 = E(1,0) * E(2,2) - E(2,0) * E(1,2);
    const float c20 = E(2,0) * E(3,1) - E(3,0) * E(2,1), c22 = E(1,0) * E(3,1) - E(3,0) * E(1,1), c23 = E(1,0) * E(2,1) - E(2,0) * E(1,1);
    const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    const float f3_[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {E(1,0), E(0,0), E(0,0), E(0,0)}, v1[4] = {E(1,1), E(0,1), E(0,1), E(0,1)};
    const float v2[4] = {E(1,2), E(0,2), E(0,2), E(0,2)}, v3[4] = {E(1,3), E(0,3), E(0,3), E(0,3)};
    static const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
    M4 inv;
    for (int i = 0; i < 4; i++) {
        inv.m[i]      = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3[i] * f2[i]) * sa[i];
        inv.m[4 + i]  = ((v0[i] * f0[i] - v2[i] * f3_[i]) + v3[i] * f4[i]) * sb[i];
        inv.m[8 + i]  = ((v0[i] * f1[i] - v1[i] * f3_[i]) + v3[i] * f5[i]) * sa[i];
        inv.m[12 + i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
    }
    const float d0 = E(0,0) * inv.m[0], d1 = E(0,1) * inv.m[4], d2 = E(0,2) * inv.m[8], d3 = E(0,3) * inv.m[12];
    const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
    for (float& x : inv.m) x = x * one_over_det;
    return inv;
}

M4 inverse_transpose(const M4& a) {
    const float s00 = E(2,2) * E(3,3) - E(3,2) * E(2,3), s01 = E(2,1) * E(3,3) - E(3,1) * E(2,3), s02 = E(2,1) * E(3,2) - E(3,1) * E(2,2);
    const float s03 = E(2,0) * E(3,3) - E(3,0) * E(2,3), s04 = E(2,0) * E(3,2) - E(3,0) * E(2,2), s05 = E(2,0) * E(3,1) - E(3,0) * E(2,1);
    const float s06 = E(1,2) * E(3,3) - E(3,2) * E(1,3), s07 = E(1,1) * E(3,3) - E(3,1) * E(1,3), s08 = E(1,1) * E(3,2) - E(3,1) * E(1,2);
    const float s09 = E(1,0) * E(3,3) - E(3,0) * E(1,3), s10 = E(1,0) * E(3,2) - E(3,0) * E(1,2), s11 = E(1,1) * E(3,3) - E(3,1) * E(1,3);
    const float s12 = E(1,0) * E(3,1) - E(3,0) * E(1,1), s13 = E(1,2) * E(2,3) - E(2,2) * E(1,3), s14 = E(1,1) * E(2,3) - E(2,1) * E(1,3);
    const float s15 = E(1,1) * E(2,2) - E(2,1) * E(1,2), s16 = E(1,0) * E(2,3) - E(2,0) * E(1,3), s17 = E(1,0) * E(2,2) - E(2,0) * E(1,2);
    const float s18 = E(1,0) * E(2,1) - E(2,0) * E(1,1);
    M4 r;
    r.m[0]  = + ((E(1,1) * s00 - E(1,2) * s01) + E(1,3) * s02);
    r.m[1]  = - ((E(1,0) * s00 - E(1,2) * s03) + E(1,3) * s04);
    r.m[2]  = + ((E(1,0) * s01 - E(1,1) * s03) + E(1,3) * s05);
    r.m[3]  = - ((E(1,0) * s02 - E(1,1) * s04) + E(1,2) * s05);
    r.m[4]  = - ((E(0,1) * s00 - E(0,2) * s01) + E(0,3) * s02);
    r.m[5]  = + ((E(0,0) * s00 - E(0,2) * s03) + E(0,3) * s04);
    r.m[6]  = - ((E(0,0) * s01 - E(0,1) * s03) + E(0,3) * s05);
    r.m[7]  = + ((E(0,0) * s02 - E(0,1) * s04) + E(0,2) * s05);
    r.m[8]  = + ((E(0,1) * s06 - E(0,2) * s07) + E(0,3) * s08);
    r.m[9]  = - ((E(0,0) * s06 - E(0,2) * s09) + E(0,3) * s10);
    r.m[10] = + ((E(0,0) * s11 - E(0,1) * s09) + E(0,3) * s12);
    r.m[11] = - ((E(0,0) * s08 - E(0,1) * s10) + E(0,2) * s12);
    r.m[12] = - ((E(0,1) * s13 - E(0,2) * s14) + E(0,3) * s15);
    r.m[13] = + ((E(0,0) * s13 - E(0,2) * s16) + E(0,3) * s17);
    r.m[14] = - ((E(0,0) * s14 - E(0,1) * s16) + E(0,3) * s18);
    r.m[15] = + ((E(0,0) * s15 - E(0,1) * s17) + E(0,2) * s18);
    const float det = ((+ E(0,0) * r.m[0] + E(0,1) * r.m[1]) + E(0,2) * r.m[2]) + E(0,3) * r.m[3];
    for (float& x : r.m) x = x / det;
    return r;
}
#undef E

M4 build_transform(const float* t, const float* rdeg, const float* s) {   // utilities.cpp:45-52
    const M4 I = identity();
    const M4 T = translate(I, t);
    M4 R = rotate(I, rdeg[0] * (float)kPi / 180, F3(1, 0, 0));
    R = mul(R, rotate(I, rdeg[1] * (float)kPi / 180, F3(0, 1, 0)));
    R = mul(R, rotate(I, rdeg[2] * (float)kPi / 180, F3(0, 0, 1)));
    const M4 S = scale(I, s);
    return mul(mul(T, R), S);
}

// utilityCore::safeGetline / tokenizeString (utilities.cpp:54-92): whitespace tokens; \n, \r\n and \r line ends
bool get_line(std::istream& is, std::string& t) {
    t.clear();
    bool any = false;
    for (;;) {
        const int c = is.get();
        if (c == EOF) { return any || !t.empty(); }
        any = true;
        if (c == '\n') return true;
        if (c == '\r') { if (is.peek() == '\n') is.get(); return true; }
        t += (char)c;
    }
}
std::vector<std::string> tokens(const std::string& s) {
    std::istringstream ss(s);
    std::vector<std::string> r;
    std::string w;
    while (ss >> w) r.push_back(w);
    return r;
}
inline float tof(const std::string& s) { return (float)atof(s.c_str()); }

void read_vec3(const std::vector<std::string>& t, float* dst) {
    for (int i = 0; i < 3; i++) dst[i] = (size_t)(i + 1) < t.size() ? tof(t[i + 1]) : 0.0f;
}

void camera_finish(aipt_scene* s) {                                          // scene.cpp:143-152
    aipt_camera& c = s->cam;
    const float yscaled = std::tan(s->fovy * (kPi / 180));
    const float xscaled = (yscaled * c.resolution[0]) / c.resolution[1];
    const float fovx = (std::atan(xscaled) * 180) / kPi;
    c.fov[0] = fovx; c.fov[1] = s->fovy;
    c.pixelLength[0] = 2 * xscaled / (float)c.resolution[0];
    c.pixelLength[1] = 2 * yscaled / (float)c.resolution[1];
    const f3 v = normalize3(sub3(F3(c.lookAt[0], c.lookAt[1], c.lookAt[2]), F3(c.position[0], c.position[1], c.position[2])));
    c.view[0] = v.x; c.view[1] = v.y; c.view[2] = v.z;
}

// Minimal Wavefront OBJ reader with tinyobjloader's defaults the reference relies on (triangulate = true,
// tiny_obj_loader.h:566): v / vn records, faces "f a//c", "f a/b/c", "f a" (1-based, negative = relative), polygons
// fanned into triangles.  Faces without normals get the geometric normal (the reference would read out of bounds).
int load_obj(aipt_scene* s, const std::string& path, int materialid, const M4& xf, unsigned flags, std::string& err) {
    std::ifstream in(path);
    if (!in.is_open()) { err = "cannot open mesh file " + path; return AIPT_E_IO; }
    std::vector<f3> vs, ns;
    struct Idx { int v, n; };
    std::string line;
    auto fix = [](int i, size_t n) { return i > 0 ? i - 1 : (i < 0 ? (int)n + i : -1); };
    while (get_line(in, line)) {
        if (line.size() < 2) continue;
        if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t')) {
            const auto t = tokens(line);
            if (t.size() >= 4) vs.push_back(F3(tof(t[1]), tof(t[2]), tof(t[3])));
        } else if (line[0] == 'v' && line[1] == 'n') {
            const auto t = tokens(line);
            if (t.size() >= 4) ns.push_back(F3(tof(t[1]), tof(t[2]), tof(t[3])));
        } else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t')) {
            const auto t = tokens(line);
            std::vector<Idx> poly;
            for (size_t k = 1; k < t.size(); k++) {
                Idx id{-1, -1};
                const std::string& w = t[k];
                const size_t s1 = w.find('/');
                id.v = fix(atoi(w.substr(0, s1).c_str()), vs.size());
                if (s1 != std::string::npos) {
                    const size_t s2 = w.find('/', s1 + 1);
                    if (s2 != std::string::npos && s2 + 1 < w.size()) id.n = fix(atoi(w.substr(s2 + 1).c_str()), ns.size());
                }
                if (id.v < 0 || id.v >= (int)vs.size()) { err = "mesh face references a missing vertex"; return AIPT_E_FORMAT; }
                if (id.n >= (int)ns.size()) { err = "mesh face references a missing normal"; return AIPT_E_FORMAT; }
                poly.push_back(id);
            }
            for (size_t k = 2; k < poly.size(); k++) {
                const Idx tri[3] = {poly[0], poly[k - 1], poly[k]};
                aipt_face f{};
                f3 p[3];
                for (int v = 0; v < 3; v++) {
                    const f3 q = vs[tri[v].v];
                    // transform * vec4(v, 1): GLM mat*vec order (type_mat4x4.inl:618-629)
                    p[v].x = (xf.m[0] * q.x + xf.m[4] * q.y) + (xf.m[8] * q.z + xf.m[12] * 1.0f);
                    p[v].y = (xf.m[1] * q.x + xf.m[5] * q.y) + (xf.m[9] * q.z + xf.m[13] * 1.0f);
                    p[v].z = (xf.m[2] * q.x + xf.m[6] * q.y) + (xf.m[10] * q.z + xf.m[14] * 1.0f);
                    f.v[v][0] = p[v].x; f.v[v][1] = p[v].y; f.v[v][2] = p[v].z;
                    const float pos[3] = {p[v].x, p[v].y, p[v].z};
                    for (int a = 0; a < 3; a++) {                                // update_mesh_box (scene.h:27-42)
                        if (s->box.lb[a] > pos[a]) s->box.lb[a] = pos[a];
                        if (s->box.ub[a] < pos[a]) s->box.ub[a] = pos[a];
                    }
                }
                const bool have_n = tri[0].n >= 0 && tri[1].n >= 0 && tri[2].n >= 0;
                const f3 gn = normalize3(cross3(sub3(p[1], p[0]), sub3(p[2], p[0])));
                for (int v = 0; v < 3; v++) {
                    const f3 n = have_n ? normalize3(ns[tri[v].n]) : gn;
                    f.n[v][0] = n.x; f.n[v][1] = n.y; f.n[v][2] = n.z;
                }
                if (flags & AIPT_SCENE_RECOMPUTE_NORMALS) {
                    // RECOMPUTE_NORMALS true (scene.cpp:9, 198-204, 310-311): normalize(cross(p2 - p0, p1 - p0)) goes to n[0] and n[1];
                    // the reference's statement assigns n[1] twice, so n[2] keeps the zero vector GLM's default constructor gave it
                    const f3 rn = normalize3(cross3(sub3(p[2], p[0]), sub3(p[1], p[0])));
                    f.n[0][0] = f.n[1][0] = rn.x; f.n[0][1] = f.n[1][1] = rn.y; f.n[0][2] = f.n[1][2] = rn.z;
                    f.n[2][0] = f.n[2][1] = f.n[2][2] = 0.0f;
                }
                f.materialid = materialid;
                s->faces.push_back(f);
            }
        }
    }
    return AIPT_OK;
}

int seterr(char* err, size_t errlen, int code, const std::string& msg) {
    if (err && errlen) { strncpy(err, msg.c_str(), errlen - 1); err[errlen - 1] = 0; }
    aipt::set_global_error(msg.c_str());
    return code;
}

}  // namespace

extern "C" {

void aipt_camera_orbit(aipt_camera* cam, float zoom, float phi, float theta) {   // main.cpp:122-140
    f3 cp;
    cp.x = zoom * std::sin(phi) * std::sin(theta);
    cp.y = zoom * std::cos(theta);
    cp.z = zoom * std::cos(phi) * std::sin(theta);
    const f3 nv = normalize3(cp);
    const f3 v = F3(-nv.x, -nv.y, -nv.z);
    const f3 r = cross3(v, F3(0, 1, 0));
    const f3 up = cross3(r, v);
    cam->view[0] = v.x; cam->view[1] = v.y; cam->view[2] = v.z;
    cam->up[0] = up.x; cam->up[1] = up.y; cam->up[2] = up.z;
    cam->right[0] = r.x; cam->right[1] = r.y; cam->right[2] = r.z;
    cam->position[0] = cp.x + cam->lookAt[0];
    cam->position[1] = cp.y + cam->lookAt[1];
    cam->position[2] = cp.z + cam->lookAt[2];
}

void aipt_camera_orbit_params(const aipt_camera* cam, float* zoom, float* phi, float* theta) {   // main.cpp:66-78
    const f3 view = F3(cam->view[0], cam->view[1], cam->view[2]);
    const f3 viewXZ = F3(view.x, 0.0f, view.z), viewZY = F3(0.0f, view.y, view.z);
    if (phi) *phi = std::acos(dot3(normalize3(viewXZ), F3(0, 0, -1)));
    if (theta) *theta = std::acos(dot3(normalize3(viewZY), F3(0, 1, 0)));
    const f3 d = sub3(F3(cam->position[0], cam->position[1], cam->position[2]), F3(cam->lookAt[0], cam->lookAt[1], cam->lookAt[2]));
    if (zoom) *zoom = std::sqrt(dot3(d, d));
}

void aipt_geom_build(aipt_geom* g) {                                             // scene.cpp:92-95
    const M4 t = build_transform(g->translation, g->rotation, g->scale);
    const M4 ti = inverse(t), tit = inverse_transpose(t);
    memcpy(g->transform, t.m, 64);
    memcpy(g->inverseTransform, ti.m, 64);
    memcpy(g->invTranspose, tit.m, 64);
}

int aipt_scene_load(const char* path, aipt_scene** out, char* err, size_t errlen) {
    return aipt_scene_load_ex(path, 0u, out, err, errlen);
}

int aipt_scene_load_ex(const char* path, unsigned flags, aipt_scene** out, char* err, size_t errlen) {
    if (!path || !out) return seterr(err, errlen, AIPT_E_INVALID, "aipt_scene_load: NULL argument");
    if (flags & ~AIPT_SCENE_RECOMPUTE_NORMALS) return seterr(err, errlen, AIPT_E_INVALID, "aipt_scene_load_ex: unknown flag");
    *out = nullptr;
    std::ifstream in(path);
    if (!in.is_open()) return seterr(err, errlen, AIPT_E_IO, std::string("cannot open scene file ") + path);
    std::unique_ptr<aipt_scene> s(new aipt_scene());
    {
        std::string p(path);
        const size_t k = p.find_last_of('/');
        s->dir = k == std::string::npos ? "." : p.substr(0, k);
    }
    bool have_camera = false;
    std::string line;
    while (get_line(in, line)) {
        const auto tk = tokens(line);
        if (tk.empty()) continue;
        if (tk[0] == "MATERIAL") {                                               // loadMaterial :161-196
            if (tk.size() < 2 || atoi(tk[1].c_str()) != (int)s->materials.size())
                return seterr(err, errlen, AIPT_E_FORMAT, "MATERIAL id does not match the number of materials so far");
            aipt_material m{};
            for (int i = 0; i < 7; i++) {
                if (!get_line(in, line)) break;
                const auto t = tokens(line);
                if (t.empty()) continue;
                if (t[0] == "RGB") read_vec3(t, m.color);
                else if (t[0] == "SPECEX" && t.size() > 1) m.specular_exponent = tof(t[1]);
                else if (t[0] == "SPECRGB") read_vec3(t, m.specular_color);
                else if (t[0] == "REFL" && t.size() > 1) m.hasReflective = tof(t[1]);
                else if (t[0] == "REFR" && t.size() > 1) m.hasRefractive = tof(t[1]);
                else if (t[0] == "REFRIOR" && t.size() > 1) m.indexOfRefraction = tof(t[1]);
                else if (t[0] == "EMITTANCE" && t.size() > 1) m.emittance = tof(t[1]);
            }
            s->materials.push_back(m);
        } else if (tk[0] == "OBJECT") {                                          // loadGeom :44-100
            if (tk.size() < 2 || atoi(tk[1].c_str()) != (int)s->geoms.size())
                return seterr(err, errlen, AIPT_E_FORMAT, "OBJECT id does not match the number of objects so far");
            aipt_geom g{};
            get_line(in, line);
            {
                const auto t = tokens(line);
                const std::string kind = t.empty() ? "" : t[0];
                if (kind == "sphere") g.type = AIPT_GEOM_SPHERE;
                else if (kind == "cube") g.type = AIPT_GEOM_CUBE;
                else return seterr(err, errlen, AIPT_E_FORMAT, "OBJECT type must be sphere or cube, got '" + kind + "'");
            }
            get_line(in, line);
            {
                const auto t = tokens(line);
                if (t.size() < 2) return seterr(err, errlen, AIPT_E_FORMAT, "OBJECT: missing material line");
                g.materialid = atoi(t[1].c_str());
            }
            while (get_line(in, line)) {
                const auto t = tokens(line);
                if (t.empty()) break;
                if (t[0] == "TRANS") read_vec3(t, g.translation);
                else if (t[0] == "ROTAT") read_vec3(t, g.rotation);
                else if (t[0] == "SCALE") read_vec3(t, g.scale);
                else if (t[0] == "VEL") read_vec3(t, g.vel);
            }
            aipt_geom_build(&g);
            s->geoms.push_back(g);
        } else if (tk[0] == "CAMERA") {                                          // loadCamera :102-159
            for (int i = 0; i < 5; i++) {
                if (!get_line(in, line)) break;
                const auto t = tokens(line);
                if (t.empty()) continue;
                if (t[0] == "RES" && t.size() > 2) { s->cam.resolution[0] = atoi(t[1].c_str()); s->cam.resolution[1] = atoi(t[2].c_str()); }
                else if (t[0] == "FOVY" && t.size() > 1) s->fovy = tof(t[1]);
                else if (t[0] == "ITERATIONS" && t.size() > 1) s->iterations = atoi(t[1].c_str());
                else if (t[0] == "DEPTH" && t.size() > 1) s->depth = atoi(t[1].c_str());
                else if (t[0] == "FILE" && t.size() > 1) s->image_name = t[1];
            }
            while (get_line(in, line)) {
                const auto t = tokens(line);
                if (t.empty()) break;
                if (t[0] == "EYE") read_vec3(t, s->cam.position);
                else if (t[0] == "LOOKAT") read_vec3(t, s->cam.lookAt);
                else if (t[0] == "UP") read_vec3(t, s->cam.up);
            }
            have_camera = true;
        } else if (tk[0] == "MESH") {                                            // loadObj :206-320
            if (tk.size() < 2 || atoi(tk[1].c_str()) != 0 || !s->faces.empty())
                return seterr(err, errlen, AIPT_E_FORMAT, "only one MESH (id 0) is supported, as in the reference");
            for (int a = 0; a < 3; a++) { s->box.lb[a] = FLT_MAX; s->box.ub[a] = FLT_MIN; }
            std::string mesh_path;
            get_line(in, line);
            { const auto t = tokens(line); if (t.size() > 1 && t[0] == "PATH") mesh_path = t[1]; }
            int materialid = 0;
            get_line(in, line);
            { const auto t = tokens(line); if (t.size() > 1) materialid = atoi(t[1].c_str()); }
            float tr[3] = {0, 0, 0}, ro[3] = {0, 0, 0}, sc[3] = {0, 0, 0};
            while (get_line(in, line)) {
                const auto t = tokens(line);
                if (t.empty()) break;
                if (t[0] == "TRANS") read_vec3(t, tr);
                else if (t[0] == "ROTAT") read_vec3(t, ro);
                else if (t[0] == "SCALE") read_vec3(t, sc);
            }
            const M4 xf = build_transform(tr, ro, sc);
            std::string e;
            std::string p1 = mesh_path;
            if (!mesh_path.empty() && mesh_path[0] != '/') {
                std::ifstream probe(s->dir + "/" + mesh_path);
                if (probe.is_open()) p1 = s->dir + "/" + mesh_path;
            }
            const int rc = load_obj(s.get(), p1, materialid, xf, flags, e);
            if (rc) return seterr(err, errlen, rc, e);
        }
    }
    if (!have_camera) return seterr(err, errlen, AIPT_E_FORMAT, "scene has no CAMERA block");
    if (s->cam.resolution[0] <= 0 || s->cam.resolution[1] <= 0) return seterr(err, errlen, AIPT_E_FORMAT, "CAMERA: bad RES");
    for (const auto& g : s->geoms)
        if (g.materialid < 0 || g.materialid >= (int)s->materials.size())
            return seterr(err, errlen, AIPT_E_FORMAT, "OBJECT references a missing material");
    for (const auto& f : s->faces)
        if (f.materialid < 0 || f.materialid >= (int)s->materials.size())
            return seterr(err, errlen, AIPT_E_FORMAT, "MESH references a missing material");
    camera_finish(s.get());
    // first-frame camera = runCuda()'s orbit rebuild from the loaded view (camchanged starts true, main.cpp:24)
    aipt_camera_orbit_params(&s->cam, &s->zoom, &s->phi, &s->theta);
    aipt_camera_orbit(&s->cam, s->zoom, s->phi, s->theta);
    *out = s.release();
    return AIPT_OK;
}

void aipt_scene_release(aipt_scene* s) { delete s; }

int aipt_scene_set_resolution(aipt_scene* s, int width, int height) {
    if (!s || width <= 0 || height <= 0) return AIPT_E_INVALID;
    s->cam.resolution[0] = width; s->cam.resolution[1] = height;
    const float yscaled = std::tan(s->fovy * (kPi / 180));
    const float xscaled = (yscaled * width) / height;
    s->cam.fov[0] = (std::atan(xscaled) * 180) / kPi;
    s->cam.pixelLength[0] = 2 * xscaled / (float)width;
    s->cam.pixelLength[1] = 2 * yscaled / (float)height;
    return AIPT_OK;
}

int aipt_scene_info(const aipt_scene* s, int* ngeoms, int* nmaterials, int* nfaces, int* iterations, int* depth) {
    if (!s) return AIPT_E_INVALID;
    if (ngeoms) *ngeoms = (int)s->geoms.size();
    if (nmaterials) *nmaterials = (int)s->materials.size();
    if (nfaces) *nfaces = (int)s->faces.size();
    if (iterations) *iterations = s->iterations;
    if (depth) *depth = s->depth;
    return AIPT_OK;
}

const aipt_geom* aipt_scene_geoms(const aipt_scene* s) { return s && !s->geoms.empty() ? s->geoms.data() : nullptr; }
const aipt_material* aipt_scene_materials(const aipt_scene* s) { return s && !s->materials.empty() ? s->materials.data() : nullptr; }
const aipt_face* aipt_scene_faces(const aipt_scene* s) { return s && !s->faces.empty() ? s->faces.data() : nullptr; }
const aipt_aabb* aipt_scene_mesh_box(const aipt_scene* s) { return s ? &s->box : nullptr; }

int aipt_scene_camera(const aipt_scene* s, aipt_camera* cam) {
    if (!s || !cam) return AIPT_E_INVALID;
    *cam = s->cam;
    return AIPT_OK;
}

int aipt_scene_orbit_params(const aipt_scene* s, float* zoom, float* phi, float* theta) {
    if (!s) return AIPT_E_INVALID;
    if (zoom) *zoom = s->zoom;
    if (phi) *phi = s->phi;
    if (theta) *theta = s->theta;
    return AIPT_OK;
}

int aipt_scene_upload_host(aipt_ctx* ctx, const aipt_scene* s) {
    AIPT_CHECK_CTX(ctx);
    if (!s) return aipt::fail(ctx, AIPT_E_INVALID, "aipt_scene_upload_host: scene is NULL");
    return aipt_scene_upload(ctx, s->geoms.data(), (int)s->geoms.size(), s->materials.data(), (int)s->materials.size(),
                             s->faces.empty() ? nullptr : s->faces.data(), (int)s->faces.size(),
                             s->faces.empty() ? nullptr : &s->box);
}

}  // extern "C"
