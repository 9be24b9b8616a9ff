// bvh.cpp -- host-side BVH over the mesh faces (the reference has none: SURVEY F1 -- one scene AABB + a brute-force loop
// over all triangles per ray per bounce, pathtrace.cu:258-269).  The tree only ACCELERATES that loop: the traversal in
// trace.hip runs the reference's triangle test on the candidate faces and resolves equal hit distances exactly as the
// index-ordered loop does, so the nearest hit is the brute-force result bit for bit (tests/test_gpu_mesh.py).
//
// What bounds the traversal kernel on MI355X is the number of per-lane vector-memory accesses (rocprofv3: TCP_TOTAL_ACCESSES =
// 1.0 per CU clock over the kernel's duration, profiles/r02_trace_pmc_v10.json), not bytes and not ALU work, so the tree is
// laid out for FEW, WIDE accesses:
//   * 4-wide nodes of 64 bytes (four 16-byte loads per lane and step): child boxes quantised to 8 bits per plane relative to
//     the node's own box (origin + per-axis power-of-two scale), rounded outward, so a decoded box always contains the padded
//     fp32 box it stands for -- boxes only prune, they never decide a hit;
//   * leaf triangles as 48-byte records {v0, e1 = v1 - v0, e2 = v2 - v0, original index} (three 16-byte loads; the edges are
//     the same fp32 subtractions glm::intersectRayTriangle starts with, gtx/intersect.inl:44-45), in leaf order;
//   * the full 76-byte face of the winning triangle is fetched once per ray, after the walk.
// Build: deterministic top-down binned SAH over a binary tree (16 bins, all three axes, centroids; object median where SAH
// degenerates and below a depth limit), then collapsed to 4-wide by repeatedly opening the child with the largest surface
// area.  Leaves hold <= BVH_LEAF_FACES faces.  build_bvh4 also returns the exact bound of the traversal stack
// (entries per lane), which sizes the kernel's LDS stack.
#include "internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace aipt {

namespace {

struct BinNode {
    float lo[3], hi[3];        // padded box
    int left = -1, right = -1; // inner: indices into bin[]; single face: left == -1
    int first = 0, count = 0;  // faces below the node: order[first, first + count)
};
// a subtree of at most BVH_LEAF_FACES faces becomes one leaf of the wide tree
inline bool is_leaf(const BinNode& n) { return n.count <= BVH_LEAF_FACES; }

struct BuildCtx {
    const aipt_face* faces;
    std::vector<int> order;            // face indices, permuted in place
    std::vector<float> cx, cy, cz;     // centroids
    std::vector<BinNode> bin;
    float pad;
    int sah_depth = 0;
    int max_depth = 0;
};

void face_bounds(const aipt_face& f, float* lo, float* hi) {
    for (int a = 0; a < 3; a++) {
        lo[a] = std::min(f.v[0][a], std::min(f.v[1][a], f.v[2][a]));
        hi[a] = std::max(f.v[0][a], std::max(f.v[1][a], f.v[2][a]));
    }
}

float half_area(const float* lo, const float* hi) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// binary tree over faces order[begin, end); returns the index of its root in c.bin
int build_rec(BuildCtx& c, int begin, int end, int depth) {
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
    const int me = (int)c.bin.size();
    c.bin.emplace_back();
    for (int a = 0; a < 3; a++) {
        const float p = c.pad + 1e-5f * std::max(std::fabs(lo[a]), std::fabs(hi[a]));
        c.bin[me].lo[a] = lo[a] - p;
        c.bin[me].hi[a] = hi[a] + p;
    }
    if (depth > c.max_depth) c.max_depth = depth;
    const int n = end - begin;
    c.bin[me].first = begin; c.bin[me].count = n;          // faces below this node: order[first, first + count)
    if (n <= 1) return me;
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    int mid = -1;
    if (depth < c.sah_depth && n > BVH_LEAF_FACES) {
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
            if (m > begin && m < end) mid = m;
        }
    }
    if (mid < 0) {                                         // object median (also the depth-bounding fallback)
        const std::vector<float>& key = axis == 0 ? c.cx : (axis == 1 ? c.cy : c.cz);
        mid = begin + n / 2;
        std::nth_element(c.order.begin() + begin, c.order.begin() + mid, c.order.begin() + end,
                         [&](int a, int b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
    }
    const int l = build_rec(c, begin, mid, depth + 1);
    const int r = build_rec(c, mid, end, depth + 1);
    c.bin[me].left = l; c.bin[me].right = r;
    return me;
}

struct Wide {
    const BuildCtx* c;
    std::vector<Bvh4Node>* nodes;
    std::vector<int>* leaf_faces;
    int max_need = 0;
};

// power-of-two scale s = 2^(e-127) with 255 * s >= ext (e in 1..254); returns the exponent byte
unsigned scale_exp(float ext) {
    if (!(ext > 0.0f)) return 1u;
    int e;
    std::frexp(ext / 255.0f, &e);                           // ext/255 = m * 2^e, m in [0.5, 1)  ->  2^e >= ext/255
    int be = e + 127;
    if (be < 1) be = 1;
    if (be > 254) be = 254;
    return (unsigned)be;
}
float exp_scale(unsigned be) {
    const uint32_t bits = be << 23;
    float s;
    memcpy(&s, &bits, 4);
    return s;
}

// emits the 4-wide node of binary inner node b; returns {node index, stack entries a lane may need below it}
std::pair<int, int> emit_wide(Wide& w, int b) {
    const std::vector<BinNode>& bin = w.c->bin;
    int kids[4], nk = 0;
    kids[nk++] = bin[b].left; kids[nk++] = bin[b].right;
    while (nk < 4) {                                        // open the inner child with the largest surface area
        int pick = -1;
        float area = -1.0f;
        for (int k = 0; k < nk; k++) {
            const BinNode& c = bin[kids[k]];
            if (is_leaf(c)) continue;
            const float a = half_area(c.lo, c.hi);
            if (a > area) { area = a; pick = k; }
        }
        if (pick < 0) break;
        const int open = kids[pick];
        kids[pick] = bin[open].left;
        kids[nk++] = bin[open].right;
    }
    const int me = (int)w.nodes->size();
    w.nodes->emplace_back();
    // quantisation frame: the union of the (padded) child boxes
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int k = 0; k < nk; k++)
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], bin[kids[k]].lo[a]); hi[a] = std::max(hi[a], bin[kids[k]].hi[a]); }
    Bvh4Node nd;
    memset(&nd, 0, sizeof(nd));
    unsigned be[3];
    float sc[3];
    for (int a = 0; a < 3; a++) {
        nd.p[a] = lo[a];
        be[a] = scale_exp(hi[a] - lo[a]);
        while (be[a] < 254 && std::fmaf(255.0f, exp_scale(be[a]), nd.p[a]) < hi[a]) be[a]++;   // fp32 rounding of hi - lo
        sc[a] = exp_scale(be[a]);
    }
    nd.exps = be[0] | (be[1] << 8) | (be[2] << 16) | ((unsigned)nk << 24);
    int refs[4] = {BVH_EMPTY, BVH_EMPTY, BVH_EMPTY, BVH_EMPTY};
    int need_below = 0;
    for (int k = 0; k < 4; k++) {
        unsigned ql[3] = {255u, 255u, 255u}, qh[3] = {0u, 0u, 0u};   // empty slot: inverted box (never entered)
        if (k < nk) {
            const BinNode& c = bin[kids[k]];
            for (int a = 0; a < 3; a++) {
                // outward rounding, verified with the kernel's own decode arithmetic: lo' = fma(q, s, p) <= lo, hi' >= hi
                int q = (int)std::floor((c.lo[a] - nd.p[a]) / sc[a]);
                q = q < 0 ? 0 : (q > 255 ? 255 : q);
                while (q > 0 && std::fmaf((float)q, sc[a], nd.p[a]) > c.lo[a]) q--;
                ql[a] = (unsigned)q;
                q = (int)std::ceil((c.hi[a] - nd.p[a]) / sc[a]);
                q = q < 0 ? 0 : (q > 255 ? 255 : q);
                while (q < 255 && std::fmaf((float)q, sc[a], nd.p[a]) < c.hi[a]) q++;
                qh[a] = (unsigned)q;
            }
        }
        for (int a = 0; a < 3; a++) {
            nd.qlo[a] |= ql[a] << (8 * k);
            nd.qhi[a] |= qh[a] << (8 * k);
        }
    }
    for (int k = 0; k < nk; k++) {
        const BinNode& c = bin[kids[k]];
        if (is_leaf(c)) {                                   // leaf: faces go to the leaf-ordered list
            refs[k] = -((int)w.leaf_faces->size() * 8 + c.count) - 1;
            for (int i = 0; i < c.count; i++) w.leaf_faces->push_back(w.c->order[c.first + i]);
        } else {
            const std::pair<int, int> r = emit_wide(w, kids[k]);
            refs[k] = r.first;
            need_below = std::max(need_below, r.second);
        }
    }
    for (int k = 0; k < 4; k++) nd.ref[k] = refs[k];
    (*w.nodes)[me] = nd;
    return {me, nk - 1 + need_below};                        // the other children wait on the stack while one is walked
}

}  // namespace

int build_bvh4(const aipt_face* faces, int nfaces, std::vector<Bvh4Node>& nodes, std::vector<int>& leaf_faces) {
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
    // SAH all the way down unless the traversal stack would come out deeper than BVH_MAX_STACK (the object-median levels
    // halve the face count, so a smaller SAH depth always terminates)
    int need = 0;
    for (int sah : {1 << 20, 24, 12, 0}) {
        c.sah_depth = sah; c.max_depth = 0;
        c.order.resize(nfaces);
        for (int i = 0; i < nfaces; i++) c.order[i] = i;
        c.bin.clear();
        c.bin.reserve((size_t)nfaces + 16);
        const int root = build_rec(c, 0, nfaces, 0);
        nodes.clear(); leaf_faces.clear();
        nodes.reserve((size_t)nfaces / 2 + 16);
        leaf_faces.reserve(nfaces);
        Wide w{&c, &nodes, &leaf_faces};
        if (is_leaf(c.bin[root])) {                            // the whole mesh is one leaf: a root node with one child
            Bvh4Node nd;
            memset(&nd, 0, sizeof(nd));
            unsigned be[3];
            for (int a = 0; a < 3; a++) { nd.p[a] = c.bin[root].lo[a]; be[a] = scale_exp(c.bin[root].hi[a] - c.bin[root].lo[a]); }
            nd.exps = be[0] | (be[1] << 8) | (be[2] << 16) | (1u << 24);
            for (int a = 0; a < 3; a++) { nd.qlo[a] = 0xFFFFFF00u; nd.qhi[a] = 0x000000FFu; }   // child 0 = the whole frame
            nd.ref[0] = -(0 * 8 + c.bin[root].count) - 1;
            nd.ref[1] = nd.ref[2] = nd.ref[3] = BVH_EMPTY;
            for (int i = 0; i < c.bin[root].count; i++) leaf_faces.push_back(c.order[i]);
            nodes.push_back(nd);
            need = 0;
        } else {
            need = emit_wide(w, root).second;
        }
        if (need <= BVH_MAX_STACK) break;
    }
    if (need > BVH_MAX_STACK) return -1;
    return need;
}

}  // namespace aipt
