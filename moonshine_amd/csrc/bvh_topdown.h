// bvh_topdown.h — the host half of the BVH builder (csrc/bvh_build.hip): the collapse-cost tables shared with the PLOC merge kernel and the top-down
// surface-area build over PLOC's clusters.  No HIP types: tests/shim/topdown_shim.cpp compiles it with g++ for the CPU-side tests.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <vector>
#if defined(__HIPCC__)
#define BVH_HD __host__ __device__
#else
#define BVH_HD
#endif

namespace msne {

struct Box { float lo[3]; float hi[3]; };

BVH_HD inline float box_area(const Box& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// Optimal collapse of a binary tree into 8-wide nodes (Ylitie, Karras, Laine 2017, section 3.1) for leaves of one primitive — every primitive is tested
// in the same place whatever the cut, so only node visits count:  cost(n, 1) = area(n) + spread(n, 8); spread(n, j) = min over k of cost(left, k) + cost(right, j - k);
// cost(n, j) = min(spread(n, j), cost(n, j - 1)); cost(primitive, j) = 0.
// cl / cr: the children's tables (zeros for a primitive); cn_out[k-1] = cost(n, k), split_out as in BinTree::split.
BVH_HD inline void collapse_table(const float cl[7], const float cr[7], float area, float cn_out[7], uint8_t split_out[8]) {
    float spread[9]; uint8_t arg[9];
    for (int j = 2; j <= 8; j++) {
        float best = 3.0e38f; int bk = 1;
        for (int k = 1; k < j; k++) if (k <= 7 && j - k <= 7) { const float v = cl[k - 1] + cr[j - k - 1]; if (v < best) { best = v; bk = k; } }
        spread[j] = best; arg[j] = (uint8_t)bk;
    }
    float cn[8];
    cn[1] = area + spread[8];
    split_out[0] = 0; split_out[7] = arg[8];
    for (int j = 2; j <= 7; j++) {
        if (spread[j] < cn[j - 1]) { cn[j] = spread[j]; split_out[j - 1] = arg[j]; }
        else { cn[j] = cn[j - 1]; split_out[j - 1] = 0; }
    }
    for (int k = 0; k < 7; k++) cn_out[k] = cn[k + 1];
}


struct TopCluster { uint32_t ref; Box box; float cost[7]; uint32_t count; };   // cost: the subtree's collapse table (zeros for a primitive)

// Host copy of the binary tree's tables, indexed by node id like BinTree's (the top-down stages write into it; whole ranges are uploaded afterwards).
// (storage that is NOT zero-filled when it is made: every entry is written before it is read, and for a million primitives the fill alone takes milliseconds)
template <class V> struct RawArray {
    std::unique_ptr<V[]> p; size_t n = 0;
    void alloc(size_t m) { p.reset(new V[m]); n = m; }
    void fill(V v) { for (size_t i = 0; i < n; i++) p[i] = v; }
    V& operator[](size_t i) { return p[i]; }
    const V& operator[](size_t i) const { return p[i]; }
    V* data() { return p.get(); }
};
struct HostTree {
    RawArray<uint32_t> left, right; RawArray<Box> box; RawArray<float> cost; RawArray<uint8_t> split;
    void alloc(size_t nodes) { left.alloc(nodes); right.alloc(nodes); box.alloc(nodes); cost.alloc(7 * nodes); split.alloc(8 * nodes); }
};

// One top-down build over m elements (clusters or primitives).  Node ids come from `ids` (as many as the build makes: m - 1), so that a subtree can be
// rebuilt in the ids it had; several builders may run on different threads over disjoint ids of one HostTree.  The elements are sorted once along every
// axis; a node is the same range [a, b) of the three orders, and a split partitions the other two orders stably, which keeps them sorted: O(m log m).
struct TopDown {
    HostTree& T; const uint32_t* ids; uint32_t used = 0, deepest = 0;   // deepest: recursion depth reached (tests)
    const TopCluster* cl = nullptr;
    std::vector<uint32_t> ord[3], tmp, count_r; std::vector<uint8_t> right_side; std::vector<float> area_r;
    struct Sub { uint32_t ref; Box box; float cost[7]; uint32_t count; };
    TopDown(HostTree& t, const uint32_t* ids_) : T(t), ids(ids_) {}
    static void grow(Box& b, const Box& o) { for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); } }
    Sub run(const TopCluster* elements, uint32_t m) {
        cl = elements;
        tmp.resize(m); count_r.resize(m); right_side.resize(m); area_r.resize(m);
        std::vector<float> key(m);
        for (int axis = 0; axis < 3; axis++) {
            ord[axis].resize(m);
            for (uint32_t i = 0; i < m; i++) { ord[axis][i] = i; key[i] = cl[i].box.lo[axis] + cl[i].box.hi[axis]; }
            std::sort(ord[axis].begin(), ord[axis].end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y] || (key[x] == key[y] && x < y); });
        }
        return build(0, m, 0);
    }
    Sub build(uint32_t a, uint32_t b, uint32_t depth) {
        deepest = std::max(deepest, depth);
        if (b - a == 1) { const TopCluster& c = cl[ord[0][a]]; Sub r; r.ref = c.ref; r.box = c.box; r.count = c.count; for (int k = 0; k < 7; k++) r.cost[k] = c.cost[k]; return r; }
        // equal costs (coincident boxes): the more even split.  Below MAX_SWEEP_DEPTH levels of lopsided splits the rest is halved along axis 0 — the
        // recursion stays shallow whatever the input
        constexpr uint32_t MAX_SWEEP_DEPTH = 64;
        const uint32_t mid = a + (b - a) / 2;
        auto off = [&](uint32_t i) { return i > mid ? i - mid : mid - i; };
        double best = 1e300; int best_axis = 0; uint32_t best_at = mid;
        for (int axis = 0; axis < 3 && depth < MAX_SWEEP_DEPTH; axis++) {
            const uint32_t* o = ord[axis].data();
            Box bx; for (int k = 0; k < 3; k++) { bx.lo[k] = 3.0e38f; bx.hi[k] = -3.0e38f; }
            uint32_t cnt = 0;
            for (uint32_t i = b; i-- > a + 1;) { grow(bx, cl[o[i]].box); cnt += cl[o[i]].count; area_r[i] = box_area(bx); count_r[i] = cnt; }
            for (int k = 0; k < 3; k++) { bx.lo[k] = 3.0e38f; bx.hi[k] = -3.0e38f; }
            cnt = 0;
            for (uint32_t i = a + 1; i < b; i++) {   // elements [a, i) go left
                grow(bx, cl[o[i - 1]].box); cnt += cl[o[i - 1]].count;
                const double c = (double)box_area(bx) * cnt + (double)area_r[i] * count_r[i];
                if (c < best || (c == best && off(i) < off(best_at))) { best = c; best_axis = axis; best_at = i; }
            }
        }
        for (uint32_t i = a; i < b; i++) right_side[ord[best_axis][i]] = i >= best_at;
        for (int axis = 0; axis < 3; axis++) {
            if (axis == best_axis) continue;
            uint32_t* o = ord[axis].data();
            uint32_t nl = a, nr = 0;
            for (uint32_t i = a; i < b; i++) { const uint32_t e = o[i]; if (right_side[e]) tmp[nr++] = e; else o[nl++] = e; }
            for (uint32_t i = 0; i < nr; i++) o[nl + i] = tmp[i];
        }
        const Sub l = build(a, best_at, depth + 1);
        const Sub r = build(best_at, b, depth + 1);
        Sub o; o.box = l.box; grow(o.box, r.box); o.count = l.count + r.count;
        uint8_t sp[8];
        collapse_table(l.cost, r.cost, box_area(o.box), o.cost, sp);
        const uint32_t id = ids[used++];
        o.ref = id;
        T.left[id] = l.ref; T.right[id] = r.ref; T.box[id] = o.box;
        for (int k = 0; k < 7; k++) T.cost[7 * (size_t)id + k] = o.cost[k];
        for (int k = 0; k < 8; k++) T.split[8 * (size_t)id + k] = sp[k];
        return o;
    }
};

}  // namespace msne
