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

// ---- the rules of the top-down sweep, shared by the GPU stages (bvh_sweep.h) and this host restatement of them (tests/test_builder_host.py,
// $MSNE_TOPDOWN=host): both produce the same binary nodes bit for bit ----
// An element's box as the sweep sees it: NaN planes (triangles with NaN vertices) become the empty interval, -0 becomes +0 — min / max over such
// values are associative and commutative, so a parallel scan and a sequential loop give the same unions.
BVH_HD inline Box canon_box(const Box& r) {
    Box b;
    for (int k = 0; k < 3; k++) { b.lo[k] = (r.lo[k] < 3.0e38f ? r.lo[k] : 3.0e38f) + 0.0f; b.hi[k] = (r.hi[k] > -3.0e38f ? r.hi[k] : -3.0e38f) + 0.0f; }
    return b;
}
BVH_HD inline void grow_box(Box& b, const Box& o) {
    for (int k = 0; k < 3; k++) { b.lo[k] = o.lo[k] < b.lo[k] ? o.lo[k] : b.lo[k]; b.hi[k] = o.hi[k] > b.hi[k] ? o.hi[k] : b.hi[k]; }
}
BVH_HD inline Box empty_box() { Box b; for (int k = 0; k < 3; k++) { b.lo[k] = 3.0e38f; b.hi[k] = -3.0e38f; } return b; }
BVH_HD inline uint32_t bits_of(float f) { return __builtin_bit_cast(uint32_t, f); }
// elements are ordered along an axis by (this key, element index): twice the centre as an order-preserving integer (total even with NaN or -0 in it)
BVH_HD inline uint32_t sweep_key(const Box& canon, int axis) { const uint32_t u = bits_of(canon.lo[axis] + canon.hi[axis]); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
// One candidate split of the range [a, b) of an axis order — elements [a, i) left, [i, b) right, mid = a + (b - a) / 2 — as ONE integer; the smallest wins:
// f32 cost area(L) * count(L) + area(R) * count(R), then the more even split, then the lower axis, then the lower position.  ~0 = no candidate (cost not below 3e38).
BVH_HD inline unsigned long long split_key(float area_l, uint32_t count_l, float area_r, uint32_t count_r, uint32_t i, uint32_t mid, int axis) {
    const float c = area_l * (float)count_l + area_r * (float)count_r;
    if (!(c < 3.0e38f)) return ~0ull;
    const uint32_t off = i > mid ? i - mid : mid - i;
    return ((unsigned long long)(bits_of(c) & 0x7fffffffu) << 33) | ((unsigned long long)off << 3) | ((unsigned long long)axis << 1) | (i > mid ? 1ull : 0ull);
}
// where a range is cut: (axis, position) of the winning key; no candidate (or the sweep switched off below MAX_SWEEP_DEPTH levels): the middle of axis 0
constexpr uint32_t MAX_SWEEP_DEPTH = 64;
BVH_HD inline void split_decode(unsigned long long key, uint32_t a, uint32_t b, int& axis, uint32_t& at) {
    const uint32_t mid = a + (b - a) / 2;
    if (key == ~0ull) { axis = 0; at = mid; return; }
    axis = (int)((key >> 1) & 3ull);
    const uint32_t off = (uint32_t)((key >> 3) & 0x3fffffffull);
    at = (key & 1ull) ? mid + off : mid - off;
}

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

// One top-down build over m elements (clusters or primitives), sequentially: the reference the GPU stages are compared with.  Node ids come from `ids`
// in post-order (as many as the build makes: m - 1), so that a subtree can be rebuilt in the ids it had; several builders may run on different threads over
// disjoint ids of one HostTree.  The elements are sorted once along every axis; a node is the same range [a, b) of the three orders, and a split
// partitions the other two orders stably, which keeps them sorted: O(m log m).
struct TopDown {
    HostTree& T; const uint32_t* ids; uint32_t used = 0, deepest = 0;   // deepest: recursion depth reached (tests)
    const TopCluster* cl = nullptr;
    std::vector<Box> cbox;                                              // canonical element boxes
    std::vector<uint32_t> ord[3], tmp, count_r; std::vector<uint8_t> right_side; std::vector<float> area_r;
    struct Sub { uint32_t ref; Box box; float cost[7]; uint32_t count; };
    TopDown(HostTree& t, const uint32_t* ids_) : T(t), ids(ids_) {}
    Sub run(const TopCluster* elements, uint32_t m) {
        cl = elements;
        tmp.resize(m); count_r.resize(m); right_side.resize(m); area_r.resize(m); cbox.resize(m);
        for (uint32_t i = 0; i < m; i++) cbox[i] = canon_box(cl[i].box);
        std::vector<uint32_t> key(m);
        for (int axis = 0; axis < 3; axis++) {
            ord[axis].resize(m);
            for (uint32_t i = 0; i < m; i++) { ord[axis][i] = i; key[i] = sweep_key(cbox[i], axis); }
            std::sort(ord[axis].begin(), ord[axis].end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y] || (key[x] == key[y] && x < y); });
        }
        return build(0, m, 0);
    }
    Sub build(uint32_t a, uint32_t b, uint32_t depth) {
        deepest = std::max(deepest, depth);
        if (b - a == 1) { const uint32_t e = ord[0][a]; const TopCluster& c = cl[e]; Sub r; r.ref = c.ref; r.box = cbox[e]; r.count = c.count; for (int k = 0; k < 7; k++) r.cost[k] = c.cost[k]; return r; }
        // equal costs (coincident boxes): the more even split.  Below MAX_SWEEP_DEPTH levels of lopsided splits the rest is halved along axis 0 — the
        // recursion stays shallow whatever the input
        const uint32_t mid = a + (b - a) / 2;
        unsigned long long best = ~0ull;
        for (int axis = 0; axis < 3 && depth < MAX_SWEEP_DEPTH; axis++) {
            const uint32_t* o = ord[axis].data();
            Box bx = empty_box();
            uint32_t cnt = 0;
            for (uint32_t i = b; i-- > a + 1;) { grow_box(bx, cbox[o[i]]); cnt += cl[o[i]].count; area_r[i] = box_area(bx); count_r[i] = cnt; }
            bx = empty_box();
            cnt = 0;
            for (uint32_t i = a + 1; i < b; i++) {   // elements [a, i) go left
                grow_box(bx, cbox[o[i - 1]]); cnt += cl[o[i - 1]].count;
                const unsigned long long k = split_key(box_area(bx), cnt, area_r[i], count_r[i], i, mid, axis);
                if (k < best) best = k;
            }
        }
        int best_axis; uint32_t best_at;
        split_decode(best, a, b, best_axis, best_at);
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
        Sub o; o.box = empty_box(); grow_box(o.box, l.box); grow_box(o.box, r.box); o.count = l.count + r.count;
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
