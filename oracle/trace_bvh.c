/*
 * trace_bvh.c -- a plain CPU BVH for the oracle's mesh loop (TEST INFRASTRUCTURE ONLY, like trace_oracle.c).
 *
 * The reference has no acceleration structure: computeIntersections tests every face for every ray (pathtrace.cu:258-269,
 * SURVEY F1).  orc_pathtrace_ex reproduces that loop; with flag 256 it asks this file for the SAME answer faster, so that
 * (a) parity tests can compare the HIP path against the oracle at the benchmark's full sizes (262 144 - 524 288 faces at
 * 1280x720 - 1920x1080) and (b) bench.py's cpu_baseline can time a CPU trace that is not O(faces) per ray, as SURVEY 8d asks.
 * It is an independent construction (binary tree, object-median splits, double-precision slab test on padded boxes), not
 * the product's 4-wide quantised tree; tests/test_oracle_trace.py checks it against the exhaustive loop, bit for bit.
 *
 * Equivalence with the index-ordered loop (strict `t_min > t`, pathtrace.cu:261): the reference's triangle test runs on the
 * candidate faces; among equal distances the lowest face index wins; a face never replaces a primitive hit at an equal
 * distance.  Boxes only prune (padded by 1e-4 of the mesh diagonal, tested in double precision with "NaN = visit").
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float v[3][3]; float n[3][3]; int materialid; } orc_face;   /* = trace_oracle.c */
float orc_triangle_test(const orc_face* f, const float* ro, const float* rd, float* P, float* N);

typedef struct { float lo[3], hi[3]; int left, right, first, count; } bnode;   /* leaf: left < 0 */
typedef struct {
    const orc_face* faces; int nfaces;
    bnode* nodes; int nnodes;
    int* order;
} orc_bvh;

static orc_bvh g_bvh = {0};

static const float* g_key;
static int cmp_key(const void* a, const void* b) {
    const int x = *(const int*)a, y = *(const int*)b;
    if (g_key[x] != g_key[y]) return g_key[x] < g_key[y] ? -1 : 1;
    return x < y ? -1 : (x > y);
}

static int build(orc_bvh* b, float* cen[3], int begin, int end, float pad) {
    const int me = b->nnodes++;
    bnode* n = &b->nodes[me];
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int a = 0; a < 3; a++) { n->lo[a] = FLT_MAX; n->hi[a] = -FLT_MAX; }
    for (int i = begin; i < end; i++) {
        const orc_face* f = &b->faces[b->order[i]];
        for (int a = 0; a < 3; a++) {
            for (int k = 0; k < 3; k++) { n->lo[a] = fminf(n->lo[a], f->v[k][a]); n->hi[a] = fmaxf(n->hi[a], f->v[k][a]); }
            clo[a] = fminf(clo[a], cen[a][b->order[i]]); chi[a] = fmaxf(chi[a], cen[a][b->order[i]]);
        }
    }
    for (int a = 0; a < 3; a++) { n->lo[a] -= pad; n->hi[a] += pad; }
    n->first = begin; n->count = end - begin; n->left = n->right = -1;
    if (end - begin <= 4) return me;
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    g_key = cen[axis];
    qsort(b->order + begin, end - begin, sizeof(int), cmp_key);
    const int mid = begin + (end - begin) / 2;
    const int l = build(b, cen, begin, mid, pad);
    const int r = build(b, cen, mid, end, pad);
    b->nodes[me].left = l; b->nodes[me].right = r;
    return me;
}

void orc_bvh_release(void) {
    free(g_bvh.nodes); free(g_bvh.order);
    memset(&g_bvh, 0, sizeof(g_bvh));
}

/* Build (or keep) the tree for this face array; single-threaded, call before the parallel trace. */
void orc_bvh_prepare(const orc_face* faces, int nfaces) {
    if (g_bvh.faces == faces && g_bvh.nfaces == nfaces && g_bvh.nodes) return;
    orc_bvh_release();
    if (nfaces <= 0) return;
    g_bvh.faces = faces; g_bvh.nfaces = nfaces;
    g_bvh.nodes = (bnode*)malloc(sizeof(bnode) * (size_t)(2 * nfaces + 2));
    g_bvh.order = (int*)malloc(sizeof(int) * (size_t)nfaces);
    float* cen[3];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int a = 0; a < 3; a++) cen[a] = (float*)malloc(sizeof(float) * (size_t)nfaces);
    for (int i = 0; i < nfaces; i++) {
        g_bvh.order[i] = i;
        for (int a = 0; a < 3; a++) {
            cen[a][i] = (faces[i].v[0][a] + faces[i].v[1][a] + faces[i].v[2][a]) * (1.0f / 3.0f);
            for (int k = 0; k < 3; k++) { lo[a] = fminf(lo[a], faces[i].v[k][a]); hi[a] = fmaxf(hi[a], faces[i].v[k][a]); }
        }
    }
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const float pad = (float)(1e-4 * sqrt(dx * dx + dy * dy + dz * dz) + 1e-6);
    build(&g_bvh, cen, 0, nfaces, pad);
    for (int a = 0; a < 3; a++) free(cen[a]);
}

/* Nearest face hit with the loop's semantics: *t_min / *best are updated exactly as the index-ordered loop over all faces
 * would leave them (best = face index or -1 when no face beats the incoming t_min, which may come from a primitive). */
void orc_bvh_nearest(const orc_face* faces, int nfaces, const float* ro, const float* rd, float* t_min, int* best,
                     float* P, float* N) {
    if (g_bvh.faces != faces || g_bvh.nfaces != nfaces || !g_bvh.nodes) { *best = -2; return; }   /* not prepared */
    const double id[3] = {1.0 / (double)rd[0], 1.0 / (double)rd[1], 1.0 / (double)rd[2]};
    int stack[128], sp = 0;
    stack[sp++] = 0;
    *best = -1;
    float tp[3], tn[3];
    while (sp) {
        const bnode* n = &g_bvh.nodes[stack[--sp]];
        double tn0 = -DBL_MAX, tf0 = DBL_MAX;
        int visit = 1;
        for (int a = 0; a < 3; a++) {
            const double t1 = ((double)n->lo[a] - (double)ro[a]) * id[a], t2 = ((double)n->hi[a] - (double)ro[a]) * id[a];
            if (t1 != t1 || t2 != t2) continue;                       /* NaN (0 * inf): no constraint from this axis */
            const double a0 = t1 < t2 ? t1 : t2, a1 = t1 < t2 ? t2 : t1;
            if (a0 > tn0) tn0 = a0;
            if (a1 < tf0) tf0 = a1;
        }
        if (tf0 < 0.0 || tn0 > tf0 || tn0 > (double)*t_min) visit = 0;
        if (!visit) continue;
        if (n->left < 0) {
            for (int i = n->first; i < n->first + n->count; i++) {
                const int fi = g_bvh.order[i];
                const float t = orc_triangle_test(&faces[fi], ro, rd, tp, tn);
                if (t > 0.0f && (*t_min > t || (*t_min == t && *best >= 0 && fi < *best))) {
                    *t_min = t; *best = fi;
                    memcpy(P, tp, 12); memcpy(N, tn, 12);
                }
            }
        } else {
            if (sp + 2 > 128) { *best = -2; return; }
            stack[sp++] = n->left; stack[sp++] = n->right;
        }
    }
}
