// bvh.cpp -- host-side BVH over the mesh faces (the reference has none: SURVEY F1 -- one scene AABB + a brute-force loop
// over all triangles per ray per bounce, pathtrace.cu:258-269).  The tree only ACCELERATES that loop: the traversal in
// trace.hip runs the reference's triangle test on the candidate faces and resolves equal hit distances exactly as the
// index-ordered loop does, so the nearest hit is the brute-force result bit for bit (tests/test_gpu_trace.py).
//
// Layout: nodes in depth-first order with a skip link ("threaded" BVH): visiting node i, a miss or a finished leaf jumps
// to skip[i], a hit on an inner node continues at i+1.  No stack, fixed traversal order, 32-byte nodes.
// Build: deterministic top-down object-median split along the largest axis of the centroid bounds (ties broken by face
// index), leaves of <= 4 faces.  Node boxes are padded so that the fp32 slab test can never reject a box whose
// triangle the exact test would hit.
#include "internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace aipt {

struct BuildCtx {
    const aipt_face* faces;
    std::vector<int> order;            // face indices, permuted in place
    std::vector<float> cx, cy, cz;     // centroids
    std::vector<BvhNode> nodes;
    std::vector<int> leaf_faces;       // face indices in leaf order
    float pad;
};

static void face_bounds(const aipt_face& f, float* lo, float* hi) {
    for (int a = 0; a < 3; a++) {
        lo[a] = std::min(f.v[0][a], std::min(f.v[1][a], f.v[2][a]));
        hi[a] = std::max(f.v[0][a], std::max(f.v[1][a], f.v[2][a]));
    }
}

static int build_rec(BuildCtx& c, int begin, int end) {
    const int me = (int)c.nodes.size();
    c.nodes.emplace_back();
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = begin; i < end; i++) {
        const int fi = c.order[i];
        float l[3], h[3];
        face_bounds(c.faces[fi], l, h);
        const float cen[3] = {c.cx[fi], c.cy[fi], c.cz[fi]};
        for (int a = 0; a < 3; a++) {
            lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]);
            clo[a] = std::min(clo[a], cen[a]); chi[a] = std::max(chi[a], cen[a]);
        }
    }
    for (int a = 0; a < 3; a++) {
        const float p = c.pad + 1e-5f * std::max(std::fabs(lo[a]), std::fabs(hi[a]));
        c.nodes[me].lo[a] = lo[a] - p;
        c.nodes[me].hi[a] = hi[a] + p;
    }
    const int n = end - begin;
    if (n <= 4) {
        c.nodes[me].leaf = ((int)c.leaf_faces.size() << 3) | n;
        for (int i = begin; i < end; i++) c.leaf_faces.push_back(c.order[i]);
    } else {
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        const std::vector<float>& key = axis == 0 ? c.cx : (axis == 1 ? c.cy : c.cz);
        const int mid = begin + n / 2;
        std::nth_element(c.order.begin() + begin, c.order.begin() + mid, c.order.begin() + end,
                         [&](int a, int b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
        c.nodes[me].leaf = -1;
        build_rec(c, begin, mid);
        build_rec(c, mid, end);
    }
    c.nodes[me].skip = (int)c.nodes.size();     // first node after this subtree in DFS order
    return me;
}

void build_bvh(const aipt_face* faces, int nfaces, std::vector<BvhNode>& nodes, std::vector<int>& leaf_faces) {
    nodes.clear(); leaf_faces.clear();
    if (nfaces <= 0) return;
    BuildCtx c;
    c.faces = faces;
    c.order.resize(nfaces);
    c.cx.resize(nfaces); c.cy.resize(nfaces); c.cz.resize(nfaces);
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < nfaces; i++) {
        c.order[i] = i;
        float l[3], h[3];
        face_bounds(faces[i], l, h);
        c.cx[i] = 0.5f * (l[0] + h[0]); c.cy[i] = 0.5f * (l[1] + h[1]); c.cz[i] = 0.5f * (l[2] + h[2]);
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    c.pad = 1e-4f * std::sqrt(dx * dx + dy * dy + dz * dz) + 1e-6f;
    c.nodes.reserve(2 * (size_t)nfaces / 2 + 16);
    c.leaf_faces.reserve(nfaces);
    build_rec(c, 0, nfaces);
    nodes.swap(c.nodes);
    leaf_faces.swap(c.leaf_faces);
}

}  // namespace aipt
