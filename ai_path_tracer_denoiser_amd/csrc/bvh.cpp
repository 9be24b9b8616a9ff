// bvh.cpp -- host-side BVH over the mesh faces (the reference has none: SURVEY F1 -- one scene AABB + a brute-force loop
// over all triangles per ray per bounce, pathtrace.cu:258-269).  The tree only ACCELERATES that loop: the traversal in
// trace.hip runs the reference's triangle test on the candidate faces and resolves equal hit distances exactly as the
// index-ordered loop does, so the nearest hit is the brute-force result bit for bit (tests/test_gpu_trace.py).
//
// Layout: 64-byte nodes holding the padded boxes of BOTH children and their references (>= 0: inner node index, < 0:
// -(first_leaf_face * 8 + count) - 1), depth-first order, root = node 0.  One fetch per inner node tests two boxes; leaves
// have no node of their own; the traversal descends into the child the ray enters first and keeps the other on a short
// per-lane stack (front to back: a hit found early prunes far subtrees through t_min).
// Build: deterministic top-down binned SAH (16 bins, all three axes, centroids); object median along the largest centroid
// axis wherever the SAH split degenerates, and below a depth limit if the pure SAH tree would be deeper than the kernel's
// per-lane stack (BVH_MAX_DEPTH).  Leaves of <= 2 faces (measured: 2 beats 4 and 6).  build_bvh returns the tree depth:
// the kernel's LDS stack is sized by it.  Node boxes
// are padded so that the fp32 slab test can never reject a box whose triangle the exact test would hit.
#include "internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>

namespace aipt {

struct BuildCtx {
    const aipt_face* faces;
    std::vector<int> order;            // face indices, permuted in place
    std::vector<float> cx, cy, cz;     // centroids
    std::vector<BvhNode> nodes;
    std::vector<int> leaf_faces;       // face indices in leaf order
    float pad;
    int sah_depth = 0;
    int max_depth = 0;
    int root = 0;
};

static void face_bounds(const aipt_face& f, float* lo, float* hi) {
    for (int a = 0; a < 3; a++) {
        lo[a] = std::min(f.v[0][a], std::min(f.v[1][a], f.v[2][a]));
        hi[a] = std::max(f.v[0][a], std::max(f.v[1][a], f.v[2][a]));
    }
}

static float half_area(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// builds the subtree of faces order[begin, end), returns its reference (>= 0: node index, < 0: -(first_leaf_face * 8 +
// count) - 1) and its padded box
static int build_rec(BuildCtx& c, int begin, int end, int depth, float* box_lo, float* box_hi) {
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
        box_lo[a] = lo[a] - p;
        box_hi[a] = hi[a] + p;
    }
    const int n = end - begin;
    if (n <= BVH_LEAF_FACES) {
        const int ref = -((int)c.leaf_faces.size() * 8 + n) - 1;
        for (int i = begin; i < end; i++) c.leaf_faces.push_back(c.order[i]);
        return ref;
    }
    const int me = (int)c.nodes.size();
    c.nodes.emplace_back();
    {
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        int mid = -1;
        if (depth < c.sah_depth && n > 2 * BVH_LEAF_FACES) {
            // binned SAH: cost(split) = area(L) * |L| + area(R) * |R| over 15 candidate planes per axis
            constexpr int NB = 16;
            float best = FLT_MAX;
            int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ax++) {
                const float ext = chi[ax] - clo[ax];
                if (!(ext > 0.0f)) continue;
                const std::vector<float>& key = ax == 0 ? c.cx : (ax == 1 ? c.cy : c.cz);
                int cnt[NB] = {0};
                float blo[NB][3], bhi[NB][3];
                for (int b = 0; b < NB; b++) for (int a = 0; a < 3; a++) { blo[b][a] = FLT_MAX; bhi[b][a] = -FLT_MAX; }
                const float scale = (float)NB / ext;
                for (int i = begin; i < end; i++) {
                    const int fi = c.order[i];
                    int b = (int)((key[fi] - clo[ax]) * scale);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    float l[3], h[3];
                    face_bounds(c.faces[fi], l, h);
                    cnt[b]++;
                    for (int a = 0; a < 3; a++) { blo[b][a] = std::min(blo[b][a], l[a]); bhi[b][a] = std::max(bhi[b][a], h[a]); }
                }
                float ra[NB];                                  // ra[b]: half area of bins b..NB-1
                int rc[NB];
                {
                    float l[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, h[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                    int k = 0;
                    for (int b = NB - 1; b >= 0; b--) {
                        if (cnt[b]) for (int a = 0; a < 3; a++) { l[a] = std::min(l[a], blo[b][a]); h[a] = std::max(h[a], bhi[b][a]); }
                        k += cnt[b];
                        rc[b] = k; ra[b] = k ? half_area(l, h) : 0.0f;
                    }
                }
                float l[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, h[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
                int k = 0;
                for (int b = 0; b + 1 < NB; b++) {             // plane between bin b and b+1
                    if (cnt[b]) for (int a = 0; a < 3; a++) { l[a] = std::min(l[a], blo[b][a]); h[a] = std::max(h[a], bhi[b][a]); }
                    k += cnt[b];
                    if (k == 0 || rc[b + 1] == 0) continue;
                    const float cost = half_area(l, h) * (float)k + ra[b + 1] * (float)rc[b + 1];
                    if (cost < best) { best = cost; best_axis = ax; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const std::vector<float>& key = best_axis == 0 ? c.cx : (best_axis == 1 ? c.cy : c.cz);
                const float scale = 16.0f / (chi[best_axis] - clo[best_axis]), base = clo[best_axis];
                auto it = std::stable_partition(c.order.begin() + begin, c.order.begin() + end, [&](int fi) {
                    int b = (int)((key[fi] - base) * scale);
                    b = b < 0 ? 0 : (b >= 16 ? 15 : b);
                    return b <= best_bin;
                });
                const int m = (int)(it - c.order.begin());
                if (m > begin && m < end) { mid = m; axis = best_axis; }
            }
        }
        if (mid < 0) {                                         // object median (also the depth-bounding fallback)
            const std::vector<float>& key = axis == 0 ? c.cx : (axis == 1 ? c.cy : c.cz);
            mid = begin + n / 2;
            std::nth_element(c.order.begin() + begin, c.order.begin() + mid, c.order.begin() + end,
                             [&](int a, int b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
        }
        float l0[3], h0[3], l1[3], h1[3];
        const int r0 = build_rec(c, begin, mid, depth + 1, l0, h0);
        const int r1 = build_rec(c, mid, end, depth + 1, l1, h1);
        BvhNode& nd = c.nodes[me];
        for (int a = 0; a < 3; a++) { nd.lo0[a] = l0[a]; nd.hi0[a] = h0[a]; nd.lo1[a] = l1[a]; nd.hi1[a] = h1[a]; }
        nd.ref0 = r0; nd.ref1 = r1; nd.pad0 = nd.pad1 = 0;
        if (depth + 1 > c.max_depth) c.max_depth = depth + 1;
    }
    return me;
}

int build_bvh(const aipt_face* faces, int nfaces, std::vector<BvhNode>& nodes, std::vector<int>& leaf_faces) {
    nodes.clear(); leaf_faces.clear();
    if (nfaces <= 0) return 0;
    BuildCtx c;
    c.faces = faces;
    c.cx.resize(nfaces); c.cy.resize(nfaces); c.cz.resize(nfaces);
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < nfaces; i++) {
        float l[3], h[3];
        face_bounds(faces[i], l, h);
        c.cx[i] = 0.5f * (l[0] + h[0]); c.cy[i] = 0.5f * (l[1] + h[1]); c.cz[i] = 0.5f * (l[2] + h[2]);
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    c.pad = 1e-4f * std::sqrt(dx * dx + dy * dy + dz * dz) + 1e-6f;
    // SAH all the way down unless the tree comes out deeper than the traversal stack allows (the object-median levels
    // halve the face count, so a smaller SAH depth always terminates: depth <= sah_depth + log2(N) + 1)
    for (int sah : {1 << 20, 16, 8, 0}) {
        c.sah_depth = sah; c.max_depth = 0;
        c.order.resize(nfaces);
        for (int i = 0; i < nfaces; i++) c.order[i] = i;
        c.nodes.clear(); c.leaf_faces.clear();
        c.nodes.reserve((size_t)nfaces + 16);
        c.leaf_faces.reserve(nfaces);
        float l[3], h[3];
        c.root = build_rec(c, 0, nfaces, 0, l, h);
        if (c.root < 0) {                                      // the whole mesh is one leaf: give it a parent whose other side is empty
            BvhNode nd;
            for (int a = 0; a < 3; a++) { nd.lo0[a] = l[a]; nd.hi0[a] = h[a]; nd.lo1[a] = FLT_MAX; nd.hi1[a] = -FLT_MAX; }
            nd.ref0 = c.root; nd.ref1 = -1; nd.pad0 = nd.pad1 = 0;     // -1: leaf of 0 faces
            c.nodes.push_back(nd);
            c.root = 0;
        }
        if (c.max_depth < BVH_MAX_DEPTH) break;
    }
    if (c.max_depth >= BVH_MAX_DEPTH) return -1;
    nodes.swap(c.nodes);
    leaf_faces.swap(c.leaf_faces);
    return c.max_depth;
}

}  // namespace aipt
