/*
 * ORACLE — test infrastructure only (see orc_math.h header).
 * orc_bvh.c: the canonical CPU BVH of SURVEY.md §8(d) and the intersection contract of
 * shaders/hrtsystem/intersection.hlsl:18-46 + main.hlsl:102-118 (closest hit: instanceIndex,
 * geometryIndex, primitiveIndex, barycentrics; shadow: any hit in (0,tmax)).
 * The reference delegates this to the Vulkan driver's BLAS/TLAS (Accel.zig:94-184,484); what is
 * restated here is the *contract*: all geometry opaque, no culling, invisible instances (mask 0,
 * Accel.zig:402) skipped, ray transformed to instance space with t preserved.
 * Result determinism: hits are decided only by tri_intersect(); equal-t ties go to the smallest
 * (instance, geometry, primitive), so any BVH over the same triangles returns the same hit.
 */
#include <stdlib.h>
#include <stdio.h>
#include "orc_scene.h"

typedef struct { v3 lo, hi; } aabb;

static inline aabb aabb_empty(void) { aabb b = { { 1e30f, 1e30f, 1e30f }, { -1e30f, -1e30f, -1e30f } }; return b; }
static inline void aabb_grow(aabb *b, v3 p) {
    b->lo.x = orc_minf(b->lo.x, p.x); b->lo.y = orc_minf(b->lo.y, p.y); b->lo.z = orc_minf(b->lo.z, p.z);
    b->hi.x = orc_maxf(b->hi.x, p.x); b->hi.y = orc_maxf(b->hi.y, p.y); b->hi.z = orc_maxf(b->hi.z, p.z);
}
static inline void aabb_merge(aabb *b, const aabb *o) { aabb_grow(b, o->lo); aabb_grow(b, o->hi); }
static inline float aabb_area(const aabb *b) {
    float dx = b->hi.x - b->lo.x, dy = b->hi.y - b->lo.y, dz = b->hi.z - b->lo.z;
    return dx * dy + dy * dz + dz * dx;
}

static inline uint32_t expand_bits10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton30(float x, float y, float z) {
    x = orc_minf(orc_maxf(x * 1024.0f, 0.0f), 1023.0f);
    y = orc_minf(orc_maxf(y * 1024.0f, 0.0f), 1023.0f);
    z = orc_minf(orc_maxf(z * 1024.0f, 0.0f), 1023.0f);
    return expand_bits10((uint32_t)x) * 4u + expand_bits10((uint32_t)y) * 2u + expand_bits10((uint32_t)z);
}

typedef struct { uint32_t left, right; aabb box; uint32_t first, count; } bnode; /* count>0 => leaf */
typedef struct {
    const uint64_t *keys; const aabb *boxes; bnode *nodes; uint32_t nnodes; uint32_t leaf_max;
} bbuild;

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static uint32_t build_range(bbuild *B, uint32_t a, uint32_t b) {
    uint32_t id = B->nnodes++;
    bnode *n = &B->nodes[id];
    if (b - a <= B->leaf_max) {
        n->left = n->right = 0; n->first = a; n->count = b - a;
        aabb bx = aabb_empty();
        for (uint32_t i = a; i < b; i++) aabb_merge(&bx, &B->boxes[(uint32_t)B->keys[i]]);
        B->nodes[id].box = bx;
        return id;
    }
    uint32_t ca = (uint32_t)(B->keys[a] >> 32), cb = (uint32_t)(B->keys[b - 1] >> 32);
    uint32_t split;
    if (ca == cb) split = (a + b) / 2;
    else {
        uint32_t diff = ca ^ cb; int bit = 31; while (!((diff >> bit) & 1u)) bit--;
        uint32_t lo = a, hi = b - 1; /* first index whose `bit` is set */
        while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (((uint32_t)(B->keys[mid] >> 32) >> bit) & 1u) hi = mid; else lo = mid + 1; }
        split = lo;
    }
    uint32_t l = build_range(B, a, split);
    uint32_t r = build_range(B, split, b);
    n = &B->nodes[id];
    n->left = l; n->right = r; n->count = 0; n->first = 0;
    n->box = B->nodes[l].box; aabb_merge(&n->box, &B->nodes[r].box);
    return id;
}

static uint32_t collapse(const bnode *bn, uint32_t root, orc_wnode *wn, uint32_t *nw) {
    uint32_t id = (*nw)++;
    uint32_t ch[8]; int nch = 0;
    if (bn[root].count > 0) { ch[nch++] = root; }
    else { ch[nch++] = bn[root].left; ch[nch++] = bn[root].right; }
    while (nch < 8) {
        int best = -1; float ba = -1.0f;
        for (int i = 0; i < nch; i++) if (bn[ch[i]].count == 0) { float ar = aabb_area(&bn[ch[i]].box); if (ar > ba) { ba = ar; best = i; } }
        if (best < 0) break;
        uint32_t c = ch[best];
        ch[best] = bn[c].left; ch[nch++] = bn[c].right;
    }
    orc_wnode w; memset(&w, 0, sizeof w);
    w.nchild = (uint8_t)nch;
    for (int i = 0; i < nch; i++) {
        const bnode *c = &bn[ch[i]];
        w.lo[0][i] = c->box.lo.x; w.lo[1][i] = c->box.lo.y; w.lo[2][i] = c->box.lo.z;
        w.hi[0][i] = c->box.hi.x; w.hi[1][i] = c->box.hi.y; w.hi[2][i] = c->box.hi.z;
        if (c->count > 0) { w.child[i] = c->first; w.count[i] = (uint8_t)c->count; }
        const float g = orc_maxf(orc_maxf(c->box.hi.x - c->box.lo.x, c->box.hi.y - c->box.lo.y), c->box.hi.z - c->box.lo.z);
        w.grow[i] = (g > 0.0f && g < 3e38f) ? 1e-4f * g : 0.0f;   /* (an empty or unbounded box: no growth) — box_hit8 */
    }
    for (int i = 0; i < nch; i++) if (bn[ch[i]].count == 0) w.child[i] = collapse(bn, ch[i], wn, nw);
    wn[id] = w;
    return id;
}

/* builds the wide BVH over n boxes; returns the permutation (item i of the BVH = original index order[i]) */
static uint32_t *bvh_build_boxes(orc_bvh *out, const aabb *boxes, uint32_t n, uint32_t leaf_max) {
    memset(out, 0, sizeof *out);
    uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    if (n == 0) { out->lo = V3(0, 0, 0); out->hi = V3(0, 0, 0); return order; }
    aabb all = aabb_empty();
    for (uint32_t i = 0; i < n; i++) aabb_merge(&all, &boxes[i]);
    out->lo = all.lo; out->hi = all.hi;
    v3 ext = v3sub(all.hi, all.lo);
    float sx = ext.x > 0.0f ? 1.0f / ext.x : 0.0f, sy = ext.y > 0.0f ? 1.0f / ext.y : 0.0f, sz = ext.z > 0.0f ? 1.0f / ext.z : 0.0f;
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (uint32_t i = 0; i < n; i++) {
        v3 c = v3scale(v3add(boxes[i].lo, boxes[i].hi), 0.5f);
        uint32_t code = morton30((c.x - all.lo.x) * sx, (c.y - all.lo.y) * sy, (c.z - all.lo.z) * sz);
        keys[i] = ((uint64_t)code << 32) | i;
    }
    qsort(keys, n, sizeof(uint64_t), cmp_u64);
    bbuild B; B.keys = keys; B.boxes = boxes; B.nnodes = 0; B.leaf_max = leaf_max;
    B.nodes = (bnode *)malloc(sizeof(bnode) * (2 * (size_t)n + 1));
    uint32_t root = build_range(&B, 0, n);
    out->nodes = (orc_wnode *)malloc(sizeof(orc_wnode) * (B.nnodes));
    out->node_count = 0;
    collapse(B.nodes, root, out->nodes, &out->node_count);
    for (uint32_t i = 0; i < n; i++) order[i] = (uint32_t)keys[i];
    free(B.nodes); free(keys);
    return order;
}

void orc_bvh_free(orc_bvh *b) { free(b->nodes); free(b->tris); free(b->inst); memset(b, 0, sizeof *b); }

/* BLAS over the triangles of a geometry list (one BLAS per unique mesh list, Accel.zig:315-343) */
void orc_build_blas(OrcContext *c, orc_bvh *out, const uint32_t *mesh_ids, uint32_t ngeo) {
    uint32_t total = 0;
    for (uint32_t g = 0; g < ngeo; g++) total += c->meshes[mesh_ids[g]].index_count;
    orc_tri *tris = (orc_tri *)malloc(sizeof(orc_tri) * (total ? total : 1));
    aabb *boxes = (aabb *)calloc(total ? total : 1, sizeof(aabb));
    uint32_t k = 0;
    for (uint32_t g = 0; g < ngeo; g++) {
        const orc_mesh *m = &c->meshes[mesh_ids[g]];
        for (uint32_t p = 0; p < m->index_count; p++, k++) {
            v3 p0 = m->positions[m->indices[3 * p + 0]], p1 = m->positions[m->indices[3 * p + 1]], p2 = m->positions[m->indices[3 * p + 2]];
            tris[k].v0 = p0; tris[k].v1 = p1; tris[k].v2 = p2; tris[k].geo = g; tris[k].prim = p;
            /* a triangle with a NaN corner is inactive (VkAccelerationStructureGeometryTrianglesDataKHR: "NaN in the first component of a vertex"; the watertight test
             * rejects it anyway): its box stays empty — grown with a NaN, min / max written as comparisons would hand the NaN on or drop it depending on the order
             * of the corners, and with it the boxes of the up to three active triangles that share its leaf and of every node above them (found by
             * tests/test_gpu_parity.py::test_random_big_scenes_match_oracle: the oracle missed hits the HIP path found) */
            boxes[k] = aabb_empty();
            if (p0.x == p0.x && p0.y == p0.y && p0.z == p0.z && p1.x == p1.x && p1.y == p1.y && p1.z == p1.z && p2.x == p2.x && p2.y == p2.y && p2.z == p2.z) {
                aabb_grow(&boxes[k], p0); aabb_grow(&boxes[k], p1); aabb_grow(&boxes[k], p2);
            }
        }
    }
    uint32_t *order = bvh_build_boxes(out, boxes, total, 4);
    out->tris = (orc_tri *)malloc(sizeof(orc_tri) * (total ? total : 1));
    out->tri_count = total;
    for (uint32_t i = 0; i < total; i++) out->tris[i] = tris[order[i]];
    free(order); free(tris); free(boxes);
}

/* How far (largest world-space coordinate difference) the world-space ray can pass from an instance's triangles and still hit one of them in instance space.
 * The hit is decided on fl(W o + w) + s fl(W d) (W, w: the rounded inverse that world_to_instance holds), the culling on o + s d, and T (W p + w) + t is not p:
 *     |T q + t - p| <= |T W - I| |p| + |T w + t| + |T| g (|W| (2 |o| + |p|) + |w|)          (g: four roundings of a dot product, doubled)
 * for a point p of the ray and its instance-space twin q; the transformed corners of the root box carry g (|T| |v| + |t|) themselves.  Everything that does not
 * depend on the ray goes into the instance's box (e0, with |p| <= the instance's reach + e0); the |o| term is per ray: far_ * max|o_a| (scene_traverse).
 * An ill-conditioned transform (shear between scales 1e6 apart) or an instance a few ulps of its own coordinates wide makes these as large as the instance itself:
 * then nothing is culled, which is the contract (OrcSetExhaustiveSearch is the same search with no boxes at all; tests hold the two against each other). */
typedef struct { float e0, far_; } inst_slack;
static inst_slack instance_cull_slack(const m34 *T, const m34 *W, v3 lo, v3 hi) {
    const double g = 8.0 / 16777216.0;
    const double a[3] = { fmax(fabs((double)lo.x), fabs((double)hi.x)), fmax(fabs((double)lo.y), fabs((double)hi.y)), fmax(fabs((double)lo.z), fabs((double)hi.z)) };
    double reach = 0.0, e_fix = 0.0, r_row = 0.0, a_row = 0.0;
    for (int i = 0; i < 3; i++) {
        double rr = 0.0, ar = 0.0, tau = (double)T->m[i][3], b = 0.0, world = fabs((double)T->m[i][3]);
        for (int j = 0; j < 3; j++) {
            double r = (i == j) ? -1.0 : 0.0, aa = 0.0;
            for (int k = 0; k < 3; k++) { r += (double)T->m[i][k] * (double)W->m[k][j]; aa += fabs((double)T->m[i][k]) * fabs((double)W->m[k][j]); }
            rr += fabs(r); ar += aa;
            tau += (double)T->m[i][j] * (double)W->m[j][3]; b += fabs((double)T->m[i][j]) * fabs((double)W->m[j][3]);
            world += fabs((double)T->m[i][j]) * a[j];
        }
        reach = fmax(reach, world);
        e_fix = fmax(e_fix, fabs(tau) + g * b + g * world);
        r_row = fmax(r_row, rr); a_row = fmax(a_row, ar);
    }
    double e0 = e_fix + (r_row + g * a_row) * reach;
    e0 = e_fix + (r_row + g * a_row) * (reach + e0);
    e0 = 1.5 * (e_fix + (r_row + g * a_row) * (reach + e0));
    inst_slack s;
    s.e0 = (e0 == e0 && e0 < 1e37) ? (float)e0 * 1.000001f : 3.0e38f;
    const double f = 1.5 * 2.0 * g * a_row;
    s.far_ = (f == f && f < 1e37) ? (float)f * 1.000001f : 3.0e38f;
    return s;
}

static aabb transform_aabb(const m34 *m, v3 lo, v3 hi) {
    aabb b = aabb_empty();
    for (int i = 0; i < 8; i++) {
        v3 p = V3((i & 1) ? hi.x : lo.x, (i & 2) ? hi.y : lo.y, (i & 4) ? hi.z : lo.z);
        aabb_grow(&b, m34_mul_point(m, p));
    }
    return b;
}

void orc_build_tlas(OrcContext *c) {
    orc_bvh_free(&c->tlas);
    uint32_t n = 0; float far_ = 0.0f;
    aabb *boxes = (aabb *)malloc(sizeof(aabb) * (c->instance_count ? c->instance_count : 1));
    uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (c->instance_count ? c->instance_count : 1));
    for (uint32_t i = 0; i < c->instance_count; i++) {
        const orc_instance *in = &c->instances[i];
        if (!in->visible) continue;
        const orc_bvh *b = &c->blases[in->blas];
        if (b->tri_count == 0) continue;
        boxes[n] = transform_aabb(&in->transform, b->lo, b->hi);
        const inst_slack sl = instance_cull_slack(&in->transform, &in->world_to_instance, b->lo, b->hi);
        boxes[n].lo = v3sub(boxes[n].lo, V3(sl.e0, sl.e0, sl.e0)); boxes[n].hi = v3add(boxes[n].hi, V3(sl.e0, sl.e0, sl.e0));
        if (!(boxes[n].lo.x > -3e38f)) boxes[n].lo = V3(-3e38f, -3e38f, -3e38f);   /* (a slack that is not finite: a box nothing misses) */
        if (!(boxes[n].hi.x < 3e38f)) boxes[n].hi = V3(3e38f, 3e38f, 3e38f);
        far_ = orc_maxf(far_, sl.far_);
        ids[n++] = i;
    }
    uint32_t *order = bvh_build_boxes(&c->tlas, boxes, n, 1);
    c->tlas.inst = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    c->tlas.inst_count = n; c->tlas.cull_far = far_;
    for (uint32_t i = 0; i < n; i++) c->tlas.inst[i] = ids[order[i]];
    free(order); free(boxes); free(ids);
}

/* ---------------- intersection ---------------- */

/* THE canonical triangle test: watertight ray/triangle intersection (Woop, Benthin, Wald, JCGT 2013),
 * no culling.  Product kernels evaluate the same expression tree, so hits are bit-identical.
 * Why not Möller–Trumbore: hardware ray tracing (the reference's TraceRay) is watertight, MT is not —
 * with MT the reference's own furnace test (tests.zig:257-344, ==1 +-1e-5) fails on this mesh (a camera
 * ray slips between two silhouette triangles, enters the sphere and is killed by Russian roulette).
 * Returns 1 with t > 0 and Vulkan barycentrics (u,v) = weights of vertices 1 and 2 (world.hlsl:118). */
typedef struct { int kx, ky, kz; float Sx, Sy, Sz; } orc_rayk;
static inline float v3idx(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
static inline orc_rayk rayk_make(v3 d) {
    orc_rayk k;
    k.kz = 0;
    if (fabsf(d.y) > fabsf(d.x)) k.kz = 1;
    if (fabsf(d.z) > fabsf(v3idx(d, k.kz))) k.kz = 2;
    k.kx = k.kz + 1; if (k.kx == 3) k.kx = 0;
    k.ky = k.kx + 1; if (k.ky == 3) k.ky = 0;
    if (v3idx(d, k.kz) < 0.0f) { int t = k.kx; k.kx = k.ky; k.ky = t; }
    float dz = v3idx(d, k.kz);
    k.Sx = v3idx(d, k.kx) / dz; k.Sy = v3idx(d, k.ky) / dz; k.Sz = 1.0f / dz;
    return k;
}
static inline int tri_intersect(v3 o, const orc_rayk *k, const orc_tri *tr, float *t, float *u, float *v) {
    const v3 A = v3sub(tr->v0, o), B = v3sub(tr->v1, o), C = v3sub(tr->v2, o);
    const float Akz = v3idx(A, k->kz), Bkz = v3idx(B, k->kz), Ckz = v3idx(C, k->kz);
    const float Ax = v3idx(A, k->kx) - k->Sx * Akz, Ay = v3idx(A, k->ky) - k->Sy * Akz;
    const float Bx = v3idx(B, k->kx) - k->Sx * Bkz, By = v3idx(B, k->ky) - k->Sy * Bkz;
    const float Cx = v3idx(C, k->kx) - k->Sx * Ckz, Cy = v3idx(C, k->ky) - k->Sy * Ckz;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx; U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx; V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax; W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    const float det = U + V + W;
    if (det == 0.0f) return 0;
    const float Az = k->Sz * Akz, Bz = k->Sz * Bkz, Cz = k->Sz * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const float rcp = 1.0f / det;
    const float tt = T * rcp;
    if (!(tt > 0.0f)) return 0;
    *t = tt; *u = V * rcp; *v = W * rcp;
    return 1;
}

static inline float safe_inv(float d) { return fabsf(d) < 1e-30f ? (d < 0.0f ? -1e30f : 1e30f) : 1.0f / d; }

/* Slab test that never culls a box holding a triangle the watertight test accepts (the hit record must not depend on the
 * BVH).  Each axis interval is widened by 1e-5 of the larger of its two plane distances: that covers the rounding of
 * (plane - o) * id, keeps equal-t candidates, and — the case a slack relative to t itself misses — a ray that runs exactly
 * in a face plane of the box (direction component 0, id = +-1e30, origin on the plane: the interval is [-2e30, 0] or
 * [0, 2e30] depending on which face) stays inside.
 * The comparison with the best hit so far carries a slack of 1.5e-6 of the box's larger plane distance ACROSS THE RAY'S DOMINANT AXIS: the watertight test computes
 * t = (U*Az + V*Bz + W*Cz) / (U + V + W), a weighted mean of the distances Az, Bz, Cz (in t) from the origin to the planes through the three vertices across that axis;
 * with every operation rounded once (A - o, 1/d, the products, the sums, the quotient) its error is at most 10 ulps (6e-7) of the LARGEST of the three — not of t.
 * A ray that starts 1e-5 in front of a triangle several units wide gets a t with a relative error of 1e-2, while the slab distance of its flat box is exact to 1e-7:
 * without the slack the second of two coincident triangles in different instances was culled against the first one's t, and the tie went to whichever instance the
 * TLAS reached first instead of the smaller index (found by tests/test_gpu_parity.py::test_random_scenes_match_oracle, seed 13).  The HIP traversal bounds the same
 * quantity per ray and space instead of per box (trace.hip cull_slack): neither side may cull a triangle that could still win, then both take the same minimum.
 * The same error sits at the OTHER end of the range: a ray that starts exactly in a triangle's plane (a camera on a lattice point of tests/hull_rays.py's scenes) or
 * a few denormals off it, leaving, has a true t of 0 or -1e-42, the test computes +2e-8 and accepts — while the triangle's flat box lies behind the origin, exactly.
 * Until round 5 the exit distance was held against 0 without the slack (found when the product's film differed from this file's on two of 750 lattice scenes and
 * turned out to agree with the search that tests every triangle: tests/test_oracle.py::test_lattice_films_with_and_without_boxes).  The product's quantised planes
 * lie at least 1e-3 quantum outside their boxes, which is what covers it there. */
/* Round 6, what the slack of the RANGE tests has to look like.  The computed t is T / det with T = U*Az + V*Bz + W*Cz and U, V, W >= 0 (or all <= 0): a convex
 * combination of Az, Bz, Cz — so it lies, to a few ulps, INSIDE THE TRIANGLE'S RANGE ALONG THE DOMINANT AXIS, and that is all that is certain about it.  U, V, W are
 * differences of products; their rounding errors move the computed barycentric point around inside the triangle by eps x size / sin(angle between ray and plane), and
 * t with it: a shadow ray that leaves a 19 x 0.1 flat quad at 1.4 degrees from 8e-7 above it has a true t of -2.4e-3 and a computed one of +6.8e-4 — an occluder for
 * the search over every triangle, while the quad's flat box lay "behind the origin" by 2.6e-6 of its far distance, more than the 1.5e-6 the range tests allowed
 * (tests/test_oracle.py::test_hull_films_with_and_without_boxes, seed 6204351, found by the GPU sweep of round 6: the product takes the hit).  The error is one of
 * POSITION ALONG THE RAY'S LINE, so per axis it is worth |1/d_a|: each axis interval is widened by 1.5e-6 x (the box's larger plane distance across the dominant axis,
 * as a length: far_ / |1/d_kz|) x |1/d_a| — for the dominant axis the 1.5e-6 far_ of before, for an axis the ray barely moves along as much more as it moves less.
 * Then the plain tests: the interval is not empty, ends after 0, starts before tmax. */
/* What box_hit needs of the ray beyond (o, 1/d), once per ray and space instead of once per box: which axes are dominant (smallest |1/d|; all that tie), and every
 * axis' |1/d| over the dominant one's (>= 1; 1 where that is not a number — a direction component that is infinite or zero against another: rays that are not rays). */
typedef struct { v3 o, id; int domx, domy, domz; float rx, ry, rz; } orc_rayb;
static inline orc_rayb rayb_make(v3 o, v3 id) {
    orc_rayb r; r.o = o; r.id = id;
    const float ax = fabsf(id.x), ay = fabsf(id.y), az = fabsf(id.z), amin = orc_minf(ax, orc_minf(ay, az));
    r.domx = ax == amin; r.domy = ay == amin; r.domz = az == amin;
    r.rx = ax / amin; r.ry = ay / amin; r.rz = az / amin;
    if (!(r.rx >= 1.0f)) r.rx = 1.0f; else if (r.rx > 3e38f) r.rx = 3e38f;   /* (finite: a box at distance 0 must get a slack of 0, not infinity x 0) */
    if (!(r.ry >= 1.0f)) r.ry = 1.0f; else if (r.ry > 3e38f) r.ry = 3e38f;
    if (!(r.rz >= 1.0f)) r.rz = 1.0f; else if (r.rz > 3e38f) r.rz = 3e38f;
    return r;
}
/* All eight boxes of a node at once (the best hit does not change between the box tests of one visit): bit i of the result = box i is hit, tnear[i] its entry
 * distance.  Every box goes through the expressions above one by one — the loop only lets the compiler use vector registers for them. */
static inline unsigned box_hit8(const orc_wnode *n, const orc_rayb *r, float tmax, float pad /* every box grown by this much on every side (TLAS: instance_cull_slack's per-ray part) */, float tnear[8]) {
    const float ox = r->o.x, oy = r->o.y, oz = r->o.z, ix = r->id.x, iy = r->id.y, iz = r->id.z;
    const float fx = r->domx ? 1.0f : 0.0f, fy = r->domy ? 1.0f : 0.0f, fz = r->domz ? 1.0f : 0.0f;   /* the dominant axis has the smallest |1/d|; among equals the larger distance */
    const float px = pad * fabsf(ix), py = pad * fabsf(iy), pz = pad * fabsf(iz);
    const float sx = 1.5e-6f * r->rx, sy = 1.5e-6f * r->ry, sz = 1.5e-6f * r->rz;
    int ok[8];
    for (int i = 0; i < 8; i++) {
        const float x1 = (n->lo[0][i] - ox) * ix, x2 = (n->hi[0][i] - ox) * ix, mx = orc_maxf(fabsf(x1), fabsf(x2));
        const float y1 = (n->lo[1][i] - oy) * iy, y2 = (n->hi[1][i] - oy) * iy, my = orc_maxf(fabsf(y1), fabsf(y2));
        const float z1 = (n->lo[2][i] - oz) * iz, z2 = (n->hi[2][i] - oz) * iz, mz = orc_maxf(fabsf(z1), fabsf(z2));
        float far_ = 0.0f;   /* (m >= 0: max(0, m) = m for a dominant axis, and an axis that is not dominant contributes 0) */
        far_ = orc_maxf(far_, fx != 0.0f ? mx : 0.0f);
        far_ = orc_maxf(far_, fy != 0.0f ? my : 0.0f);
        far_ = orc_maxf(far_, fz != 0.0f ? mz : 0.0f);
        /* ... and 1e-4 of the box's own largest extent, as a length, on every side: the amplification above has no bound in what a NODE knows — a needle of a triangle
         * (15.8 x 0.028, seed 6226272: t = +3.7e-4 for a true -1.5e-4 at 24 degrees from its plane) amplifies by its aspect ratio as a grazing ray does by 1 / sin —
         * but the error is always a fraction of the TRIANGLE's size, and a triangle is no larger than a box that holds it: 1e-4 covers amplifications up to ~1600.
         * The product's builder grows its boxes by the same 1e-4 before it quantises them (bvh_build.hip box_growth). */
        const float g = n->grow[i];   /* (computed when the node is made: collapse()) */
        float e = 1e-5f * mx + px + sx * far_ + g * fabsf(ix);
        float tn = orc_minf(x1, x2) - e, tf = orc_maxf(x1, x2) + e;
        e = 1e-5f * my + py + sy * far_ + g * fabsf(iy);
        tn = orc_maxf(tn, orc_minf(y1, y2) - e); tf = orc_minf(tf, orc_maxf(y1, y2) + e);
        e = 1e-5f * mz + pz + sz * far_ + g * fabsf(iz);
        tn = orc_maxf(tn, orc_minf(z1, z2) - e); tf = orc_minf(tf, orc_maxf(z1, z2) + e);
        tnear[i] = tn;
        ok[i] = (tn <= tf) & (tf >= 0.0f) & (tn <= tmax);   /* (every axis interval carries its slack already) */
    }
    unsigned m = 0;
    for (int i = 0; i < n->nchild; i++) m |= (unsigned)(ok[i] != 0) << i;
    return m;
}

/* traverse one BLAS in instance space. any_hit: return at the first triangle with t < tmax */
static int blas_traverse(const orc_bvh *b, v3 o, v3 d, uint32_t inst, orc_hit *best, int any_hit, orc_counters *cnt) {
    if (b->node_count == 0) return 0;
    const orc_rayb rb = rayb_make(o, V3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)));
    const orc_rayk rk = rayk_make(d);
    uint32_t stack[256]; int sp = 0; int found = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const orc_wnode *n = &b->nodes[stack[--sp]];
        cnt->node_visits++; if (any_hit) cnt->shadow_node_visits++;
        float tn[8], tb[8]; int idx[8]; int nh = 0;
        const unsigned hm = box_hit8(n, &rb, best->t, 0.0f, tb);
        for (int i = 0; i < n->nchild; i++) {
            const float t = tb[i];
            if (hm >> i & 1u) {
                int j = nh++;
                while (j > 0 && tn[j - 1] < t) { tn[j] = tn[j - 1]; idx[j] = idx[j - 1]; j--; } /* far first */
                tn[j] = t; idx[j] = i;
            }
        }
        for (int k = 0; k < nh; k++) {
            int i = idx[k];
            if (n->count[i] == 0) {
                if (sp < 255) stack[sp++] = n->child[i]; else { fprintf(stderr, "orc: bvh stack overflow\n"); abort(); }
                const char *pf = (const char *)&b->nodes[n->child[i]];   /* (the tree is far larger than the caches: a visit is a chain of misses) */
                __builtin_prefetch(pf); __builtin_prefetch(pf + 64); __builtin_prefetch(pf + 128); __builtin_prefetch(pf + 192);
            } else __builtin_prefetch(&b->tris[n->child[i]]);
        }
        for (int k = nh - 1; k >= 0; k--) { /* leaves: near first */
            int i = idx[k];
            if (n->count[i] == 0) continue;
            for (uint32_t q = 0; q < n->count[i]; q++) {
                const orc_tri *tr = &b->tris[n->child[i] + q];
                float t, u, v;
                cnt->tri_tests++; if (any_hit) cnt->shadow_tri_tests++;
                if (!tri_intersect(o, &rk, tr, &t, &u, &v)) continue;
                if (any_hit) { if (t < best->t) return 1; continue; }
                int closer = t < best->t;
                if (!closer && t == best->t && best->inst != ORC_MAX_UINT) {
                    closer = inst < best->inst || (inst == best->inst && (tr->geo < best->geo || (tr->geo == best->geo && tr->prim < best->prim)));
                }
                if (closer) { best->t = t; best->u = u; best->v = v; best->inst = inst; best->geo = tr->geo; best->prim = tr->prim; found = 1; }
            }
        }
    }
    return found;
}

/* The contract itself, with no acceleration structure in the way (OrcSetExhaustiveSearch): every active triangle of the BLAS against the instance-space ray, the same
 * watertight test, the same order of preference.  What the box culls above and below must never change — tests hold the culled search against this one. */
static int blas_exhaustive(const orc_bvh *b, v3 o, v3 d, uint32_t inst, orc_hit *best, int any_hit, orc_counters *cnt) {
    const orc_rayk rk = rayk_make(d);
    int found = 0;
    for (uint32_t q = 0; q < b->tri_count; q++) {
        const orc_tri *tr = &b->tris[q];
        float t, u, v;
        cnt->tri_tests++; if (any_hit) cnt->shadow_tri_tests++;
        if (!tri_intersect(o, &rk, tr, &t, &u, &v)) continue;
        if (any_hit) { if (t < best->t) return 1; continue; }
        int closer = t < best->t;
        if (!closer && t == best->t && best->inst != ORC_MAX_UINT)
            closer = inst < best->inst || (inst == best->inst && (tr->geo < best->geo || (tr->geo == best->geo && tr->prim < best->prim)));
        if (closer) { best->t = t; best->u = u; best->v = v; best->inst = inst; best->geo = tr->geo; best->prim = tr->prim; found = 1; }
    }
    return found;
}
static int scene_exhaustive(const OrcContext *c, v3 o, v3 d, orc_hit *best, int any_hit, orc_counters *cnt) {
    int found = 0;
    for (uint32_t q = 0; q < c->tlas.inst_count; q++) {   /* the visible instances that have triangles (orc_build_tlas) */
        const uint32_t ii = c->tlas.inst[q];
        const orc_instance *in = &c->instances[ii];
        const v3 oo = m34_mul_point(&in->world_to_instance, o), dd = m34_mul_vec(&in->world_to_instance, d);
        const orc_bvh *b = &c->blases[in->blas];
        if (c->exhaustive >= 2 ? blas_exhaustive(b, oo, dd, ii, best, any_hit, cnt) : blas_traverse(b, oo, dd, ii, best, any_hit, cnt)) { found = 1; if (any_hit) return 1; }
    }
    return found;
}

static int scene_traverse(const OrcContext *c, v3 o, v3 d, float tmax, orc_hit *best, int any_hit, orc_counters *cnt) {
    best->inst = ORC_MAX_UINT; best->t = tmax; best->geo = best->prim = 0; best->u = best->v = 0.0f;
    if (c->exhaustive) return scene_exhaustive(c, o, d, best, any_hit, cnt);
    const orc_bvh *tl = &c->tlas;
    if (tl->node_count == 0) return 0;
    const orc_rayb rb = rayb_make(o, V3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)));
    const float far_pad = tl->cull_far * orc_maxf(orc_maxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));   /* instance_cull_slack: the part that grows with the origin's coordinates */
    uint32_t stack[256]; int sp = 0; int found = 0;
    stack[sp++] = 0;
    while (sp > 0) {
        const orc_wnode *n = &tl->nodes[stack[--sp]];
        cnt->node_visits++; if (any_hit) cnt->shadow_node_visits++;
        float tn[8], tb[8]; int idx[8]; int nh = 0;
        const unsigned hm = box_hit8(n, &rb, best->t, far_pad, tb);
        for (int i = 0; i < n->nchild; i++) {
            const float t = tb[i];
            if (hm >> i & 1u) {
                int j = nh++;
                while (j > 0 && tn[j - 1] < t) { tn[j] = tn[j - 1]; idx[j] = idx[j - 1]; j--; }
                tn[j] = t; idx[j] = i;
            }
        }
        for (int k = 0; k < nh; k++) { int i = idx[k]; if (n->count[i] == 0) stack[sp++] = n->child[i]; }
        for (int k = nh - 1; k >= 0; k--) {
            int i = idx[k];
            if (n->count[i] == 0) continue;
            for (uint32_t q = 0; q < n->count[i]; q++) {
                uint32_t ii = tl->inst[n->child[i] + q];
                const orc_instance *in = &c->instances[ii];
                v3 oo = m34_mul_point(&in->world_to_instance, o);
                v3 dd = m34_mul_vec(&in->world_to_instance, d);
                if (blas_traverse(&c->blases[in->blas], oo, dd, ii, best, any_hit, cnt)) { found = 1; if (any_hit) return 1; }
            }
        }
    }
    return found;
}

/* Intersection::find (intersection.hlsl:18-22): tmin = 0, tmax = ray.tmax */
int orc_closest_hit(const OrcContext *c, v3 o, v3 d, float tmax, orc_hit *h, orc_counters *cnt) {
    cnt->closest_rays++;
    return scene_traverse(c, o, d, tmax, h, 0, cnt);
}
/* ShadowIntersection::hit (intersection.hlsl:33-46) */
int orc_shadow_hit(const OrcContext *c, v3 o, v3 d, float tmax, orc_counters *cnt) {
    orc_hit h;
    cnt->shadow_rays++;
    return scene_traverse(c, o, d, tmax, &h, 1, cnt);
}
