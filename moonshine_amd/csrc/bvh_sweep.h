// bvh_sweep.h — the top-down surface-area stages of the BVH builder ON THE GPU (included by bvh_build.hip; the sequential statement of the same rules is
// TopDown in bvh_topdown.h, and both make the same binary nodes bit for bit: tests/test_gpu_parity.py::test_gpu_sweep_builds_the_hosts_nodes).
//
// PLOC stops at a few thousand clusters.  Every cluster is rebuilt over its own primitives and the clusters themselves become the top of the tree, both by a full
// sweep of the surface-area heuristic over the three axes — ALL of those builds advance together, one tree level per pass:
//   * elements = the primitives grouped by cluster (positions [0, P0)) followed by the clusters grouped by tree (positions [P0, N)); a tree node in the making is
//     a SEGMENT [a, b) of positions, the same range in three orders of the elements (sorted along x, y, z once, by radix sort);
//   * per level: k_sw_agg / k_sw_cost evaluate every split of every segment along every axis by segmented prefix / suffix scans of the boxes (256 positions per
//     workgroup, wave shuffles inside, per-tile aggregates across) and keep the best one per segment with one 64-bit atomicMin (split_key); k_sw_split marks the
//     elements that go right, names the node (ids in post-order out of the cluster's own sorted id list, as TopDown assigns them) and cuts the segment;
//     k_sw_pagg / k_sw_part partition the three orders stably (segmented counts of the left-going elements);
//   * then the boxes and collapse tables bottom-up, level by level (k_sw_up): cluster trees first, the top trees over them after.
// No host work beyond sequencing launches and reading the per-level node counts back every few levels.
#pragma once

namespace msne {

constexpr int SW_TILE = 256;                 // positions per workgroup
constexpr uint32_t SW_MAX_LEVELS = 192;      // MAX_SWEEP_DEPTH lopsided levels + log2 of the largest segment halved after them
constexpr uint32_t SW_NONE = 0xFFFFFFFFu;
constexpr uint32_t SW_SUPER = 64;           // tiles per super-tile

struct alignas(16) SwAgg { float lo[3]; float hi[3]; uint32_t cnt; uint32_t pad; };   // union of boxes + sum of primitive counts
__device__ __forceinline__ SwAgg sw_identity() { SwAgg v; for (int k = 0; k < 3; k++) { v.lo[k] = 3.0e38f; v.hi[k] = -3.0e38f; } v.cnt = 0u; v.pad = 0u; return v; }
__device__ __forceinline__ void sw_add(SwAgg& x, const SwAgg& y) {
    for (int k = 0; k < 3; k++) { x.lo[k] = y.lo[k] < x.lo[k] ? y.lo[k] : x.lo[k]; x.hi[k] = y.hi[k] > x.hi[k] ? y.hi[k] : x.hi[k]; }
    x.cnt += y.cnt;
}
__device__ __forceinline__ SwAgg sw_shfl(const SwAgg& v, int src_lane) {
    SwAgg o;
    for (int k = 0; k < 3; k++) { o.lo[k] = __shfl(v.lo[k], src_lane); o.hi[k] = __shfl(v.hi[k], src_lane); }
    o.cnt = (uint32_t)__shfl((int)v.cnt, src_lane); o.pad = 0u;
    return o;
}
__device__ __forceinline__ float sw_area(const SwAgg& v) { Box b; for (int k = 0; k < 3; k++) { b.lo[k] = v.lo[k]; b.hi[k] = v.hi[k]; } return box_area(b); }

// sum over the workgroup (256 threads); every thread gets the result.  s: 4 entries.
__device__ __forceinline__ SwAgg sw_block_reduce(SwAgg v, SwAgg* s) {
    for (int o = 32; o >= 1; o >>= 1) { const SwAgg t = sw_shfl(v, (int)((threadIdx.x & 63u) ^ (uint32_t)o)); sw_add(v, t); }
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    SwAgg r = s[0];
    for (int w = 1; w < SW_TILE / 64; w++) sw_add(r, s[w]);
    return r;
}

// everything a level's kernels read (device pointers); `cur` selects the ping-pong halves
struct SwState {
    uint32_t N, P0, ngc, node_base, ntile;      // positions, primitive positions, primitive groups (clusters), first id of the top trees, tiles
    const Box* ebox; const uint32_t* ecnt; const uint32_t* eref; const uint32_t* egrp;   // per element (= per base position): canonical box, count, ref, group
    const uint32_t* cid;                        // the clusters' PLOC node ids, grouped by cluster, ascending inside
    uint32_t* ord[2][3];                        // element at a position, per axis
    uint32_t* segA[2]; uint32_t* segB[2]; uint32_t* segR[2]; unsigned long long* best[2];   // segment start per position; end / right turns / best split per segment start
    uint8_t* tile_act[2]; uint8_t* right_side;
    SwAgg* vf[3]; SwAgg* vb[3]; uint32_t* cf[3]; // per tile and axis: aggregates of the segment that leaves the tile at its end / enters it at its start; left-goers of the former
    SwAgg* vf2[3]; SwAgg* vb2[3]; uint32_t* cf2[3]; // the same summed over SW_SUPER consecutive tiles (a segment of a million positions spans 4000 tiles: nobody walks them one by one)
    uint32_t* list[2]; uint32_t* list_count;     // node ids by kind (0 cluster trees, 1 top trees) in level order; [2] counters
    uint32_t* level_first;                       // [2][SW_MAX_LEVELS + 1]
};

__device__ __forceinline__ uint32_t sw_node_id(const SwState& S, uint32_t a, uint32_t b, uint32_t r) {
    // post-order index of the node over [a, b) inside its group's id list: (leaves before b) - (right turns) - 2; the group's list starts at (first position) - (group index)
    const uint32_t g = S.egrp[a];
    return a < S.P0 ? S.cid[b - g - r - 2u] : S.node_base + (b - S.P0) - (g - S.ngc) - r - 2u;
}

__global__ __launch_bounds__(SW_TILE) void k_sw_init(SwState S, const uint32_t* grp_first) {
    const uint32_t p = blockIdx.x * SW_TILE + threadIdx.x;
    bool act = false;
    if (p < S.N) {
        const uint32_t g = S.egrp[p], a = grp_first[g], b = grp_first[g + 1u];
        S.segA[0][p] = a;
        if (p == a) { S.segB[0][a] = b; S.segR[0][a] = 0u; S.best[0][a] = ~0ull; }
        act = b - a >= 2u;
    }
    const int any = __syncthreads_or(act ? 1 : 0);
    if (threadIdx.x == 0) S.tile_act[0][blockIdx.x] = (uint8_t)(any != 0);
}

// per tile and axis: the aggregates the NEIGHBOURING tiles need — of the segment that is still open at the tile's end (positions from its start or the tile's
// start on), and of the one that was already open at the tile's start (positions up to its end or the tile's end)
__global__ __launch_bounds__(SW_TILE) void k_sw_agg(SwState S, int cur, uint32_t level) {
    __shared__ SwAgg s_red[SW_TILE / 64];
    const uint32_t t = blockIdx.x, axis = blockIdx.y, t0 = t * SW_TILE, t1 = min(t0 + (uint32_t)SW_TILE, S.N), p = t0 + threadIdx.x;
    if (t == 0 && axis == 0 && threadIdx.x < 2) S.level_first[threadIdx.x * (SW_MAX_LEVELS + 1u) + level] = S.list_count[threadIdx.x];
    if (!S.tile_act[cur][t]) return;
    const uint32_t a_last = S.segA[cur][t1 - 1u], b_last = S.segB[cur][a_last], a_first = S.segA[cur][t0], b_first = S.segB[cur][a_first];
    const bool need_f = b_last > t1, need_b = a_first < t0;
    if (!need_f && !need_b) return;
    SwAgg v = sw_identity();
    if (p < t1) { const uint32_t e = S.ord[cur][axis][p]; const Box bx = S.ebox[e]; for (int k = 0; k < 3; k++) { v.lo[k] = bx.lo[k]; v.hi[k] = bx.hi[k]; } v.cnt = S.ecnt[e]; }
    if (need_f) { const SwAgg r = sw_block_reduce((p < t1 && p >= a_last) ? v : sw_identity(), s_red); if (threadIdx.x == 0) S.vf[axis][t] = r; }
    if (need_b) { const SwAgg r = sw_block_reduce((p < t1 && p < b_first) ? v : sw_identity(), s_red); if (threadIdx.x == 0) S.vb[axis][t] = r; }
}

// Sum of the per-tile aggregates fine[lo .. hi) by the whole workgroup: whole super-tiles inside the range come from `coarse`.  Returns this thread's share (reduce it).
__device__ __forceinline__ SwAgg sw_walk(const SwAgg* fine, const SwAgg* coarse, uint32_t lo, uint32_t hi) {
    SwAgg c = sw_identity();
    const uint32_t s0 = (lo + SW_SUPER - 1u) / SW_SUPER, s1 = hi / SW_SUPER;
    if (s0 < s1) {
        const uint32_t n0 = s0 * SW_SUPER - lo, n1 = s1 - s0, n2 = hi - s1 * SW_SUPER;
        for (uint32_t i = threadIdx.x; i < n0 + n1 + n2; i += SW_TILE)
            sw_add(c, i < n0 ? fine[lo + i] : (i < n0 + n1 ? coarse[s0 + (i - n0)] : fine[s1 * SW_SUPER + (i - n0 - n1)]));
    } else for (uint32_t i = lo + threadIdx.x; i < hi; i += SW_TILE) sw_add(c, fine[i]);
    return c;
}
__device__ __forceinline__ uint32_t sw_walk_count(const uint32_t* fine, const uint32_t* coarse, uint32_t lo, uint32_t hi) {
    uint32_t c = 0;
    const uint32_t s0 = (lo + SW_SUPER - 1u) / SW_SUPER, s1 = hi / SW_SUPER;
    if (s0 < s1) {
        const uint32_t n0 = s0 * SW_SUPER - lo, n1 = s1 - s0, n2 = hi - s1 * SW_SUPER;
        for (uint32_t i = threadIdx.x; i < n0 + n1 + n2; i += SW_TILE)
            c += i < n0 ? fine[lo + i] : (i < n0 + n1 ? coarse[s0 + (i - n0)] : fine[s1 * SW_SUPER + (i - n0 - n1)]);
    } else for (uint32_t i = lo + threadIdx.x; i < hi; i += SW_TILE) c += fine[i];
    return c;
}
// per super-tile and axis: the tiles' aggregates summed (only ever read for super-tiles that lie wholly inside one segment, where every tile's entry is this level's)
__global__ __launch_bounds__(64) void k_sw_agg2(SwState S, int cur) {
    const uint32_t axis = blockIdx.y, j = blockIdx.x * SW_SUPER + threadIdx.x;
    const bool valid = j < S.ntile;
    if (__ballot(valid && S.tile_act[cur][j]) == 0ull) return;
    SwAgg f = valid ? S.vf[axis][j] : sw_identity(), b = valid ? S.vb[axis][j] : sw_identity();
    for (int o = 32; o >= 1; o >>= 1) { const SwAgg tf = sw_shfl(f, (int)(threadIdx.x ^ (uint32_t)o)), tb = sw_shfl(b, (int)(threadIdx.x ^ (uint32_t)o)); sw_add(f, tf); sw_add(b, tb); }
    if (threadIdx.x == 0) { S.vf2[axis][blockIdx.x] = f; S.vb2[axis][blockIdx.x] = b; }
}
__global__ __launch_bounds__(64) void k_sw_pagg2(SwState S, int cur) {
    const uint32_t axis = blockIdx.y, j = blockIdx.x * SW_SUPER + threadIdx.x;
    const bool valid = j < S.ntile;
    if (__ballot(valid && S.tile_act[cur][j]) == 0ull) return;
    uint32_t c = valid ? S.cf[axis][j] : 0u;
    for (int o = 32; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
    if (threadIdx.x == 0) S.cf2[axis][blockIdx.x] = c;
}

// every split of every segment along one axis: cost from the inclusive prefix up to p and the exclusive suffix after p (split between p and p + 1)
__global__ __launch_bounds__(SW_TILE) void k_sw_cost(SwState S, int cur) {
    __shared__ SwAgg s_red[SW_TILE / 64], s_wf[SW_TILE / 64], s_wb[SW_TILE / 64], s_first[SW_TILE / 64 + 1];
    __shared__ uint32_t s_hf[SW_TILE / 64], s_hb[SW_TILE / 64];
    const uint32_t t = blockIdx.x, axis = blockIdx.y, t0 = t * SW_TILE, t1 = min(t0 + (uint32_t)SW_TILE, S.N), p = t0 + threadIdx.x;
    if (!S.tile_act[cur][t]) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool valid = p < t1;
    uint32_t a = p, b = p + 1u;
    SwAgg v = sw_identity();
    if (valid) {
        a = S.segA[cur][p]; b = S.segB[cur][a];
        const uint32_t e = S.ord[cur][axis][p]; const Box bx = S.ebox[e];
        for (int k = 0; k < 3; k++) { v.lo[k] = bx.lo[k]; v.hi[k] = bx.hi[k]; } v.cnt = S.ecnt[e];
    }
    // carries from the tiles before / after (the segment open at the tile's start / end); every thread of the tile computes the same two sums
    const uint32_t a_first = S.segA[cur][t0], a_last = S.segA[cur][t1 - 1u], b_last = S.segB[cur][a_last];
    SwAgg carry_f = sw_identity(), carry_b = sw_identity();
    if (a_first < t0) carry_f = sw_block_reduce(sw_walk(S.vf[axis], S.vf2[axis], a_first / SW_TILE, t), s_red);
    if (b_last > t1) carry_b = sw_block_reduce(sw_walk(S.vb[axis], S.vb2[axis], t + 1u, (b_last - 1u) / SW_TILE + 1u), s_red);
    // forward inclusive segmented scan inside the wave (heads: segment starts), then across the tile's waves
    SwAgg P = v; bool hf = valid ? (p == a) : true;
    for (int d = 1; d < 64; d <<= 1) {
        const SwAgg o = sw_shfl(P, (int)lane - d); const int oh = __shfl((int)hf, (int)lane - d);
        if ((int)lane >= d) { if (!hf) sw_add(P, o); hf = hf || (oh != 0); }
    }
    // backward inclusive segmented scan (heads: segment ends)
    SwAgg Sx = v; bool hb = valid ? (p == b - 1u) : true;
    for (int d = 1; d < 64; d <<= 1) {
        const SwAgg o = sw_shfl(Sx, (int)lane + d); const int oh = __shfl((int)hb, (int)lane + d);
        if ((int)lane + d < 64) { if (!hb) sw_add(Sx, o); hb = hb || (oh != 0); }
    }
    if (lane == 63u) { s_wf[wave] = P; s_hf[wave] = hf; }
    if (lane == 0u) { s_wb[wave] = Sx; s_hb[wave] = hb; }
    __syncthreads();
    if (!hf) {   // the segment started before this wave: add the open tails of the waves before, back to the one that holds its start, or the tile carry
        SwAgg c = sw_identity(); bool closed = false;
        for (int w = (int)wave - 1; w >= 0 && !closed; w--) { sw_add(c, s_wf[w]); closed = s_hf[w] != 0u; }
        if (!closed) sw_add(c, carry_f);
        sw_add(P, c);
    }
    if (!hb) {
        SwAgg c = sw_identity(); bool closed = false;
        for (int w = (int)wave + 1; w < SW_TILE / 64 && !closed; w++) { sw_add(c, s_wb[w]); closed = s_hb[w] != 0u; }
        if (!closed) sw_add(c, carry_b);
        sw_add(Sx, c);
    }
    // exclusive suffix: the inclusive one of position p + 1 (next lane, next wave's first lane, or the carry from the tiles after)
    __syncthreads();
    if (lane == 0u) s_first[wave] = Sx;
    if (threadIdx.x == 0) s_first[SW_TILE / 64] = carry_b;
    __syncthreads();
    SwAgg X = sw_shfl(Sx, (int)lane + 1);
    if (lane == 63u) X = s_first[wave + 1u];
    if (p + 1u == t1) X = carry_b;
    unsigned long long key = ~0ull;
    if (valid && b - a >= 2u && p + 1u < b) key = split_key(sw_area(P), P.cnt, sw_area(X), X.cnt, p + 1u, a + (b - a) / 2u, (int)axis);
    // one atomic per segment and wave: the minimum runs forward through the lanes of a segment, its last lane in the wave publishes it
    {
        bool h = valid ? (p == a) : true;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long ok = __shfl(key, (int)lane - d); const int oh = __shfl((int)h, (int)lane - d);
            if ((int)lane >= d) { if (!h && ok < key) key = ok; h = h || (oh != 0); }
        }
        const uint32_t a_next = (uint32_t)__shfl((int)a, (int)lane + 1);
        if (valid && key != ~0ull && (lane == 63u || a_next != a)) atomicMin(&S.best[cur][a], key);
    }
}

// cut every segment of two or more positions where its best split says: elements that go right are marked, the node is named and listed, the next level's
// segments are written to the other half of the ping-pong state
__global__ __launch_bounds__(SW_TILE) void k_sw_split(SwState S, int cur, BinTree tr) {
    const uint32_t t = blockIdx.x, p = t * SW_TILE + threadIdx.x;
    const int nxt = cur ^ 1;
    if (!S.tile_act[cur][t]) { if (threadIdx.x == 0) S.tile_act[nxt][t] = 0; return; }
    bool act = false;
    uint32_t new_id = SW_NONE, kind = 0;       // the node this thread names (threads at a segment's start)
    if (p < S.N) {
        const uint32_t a = S.segA[cur][p], b = S.segB[cur][a];
        if (b - a >= 2u) {
            const uint32_t r = S.segR[cur][a];
            int axis; uint32_t at;
            split_decode(S.best[cur][a], a, b, axis, at);
            S.right_side[S.ord[cur][axis][p]] = (uint8_t)(p >= at);
            S.segA[nxt][p] = p < at ? a : at;
            if (p == a) {
                S.segB[nxt][a] = at; S.segR[nxt][a] = r; S.best[nxt][a] = ~0ull;
                new_id = sw_node_id(S, a, b, r); kind = a < S.P0 ? 0u : 1u;
                if (at - a >= 2u) tr.left[new_id] = sw_node_id(S, a, at, r);          // (children of one element are written by k_sw_part, which sees the element land)
                if (b - at >= 2u) tr.right[new_id] = sw_node_id(S, at, b, r + 1u);
            }
            if (p == at) { S.segB[nxt][at] = b; S.segR[nxt][at] = r + 1u; S.best[nxt][at] = ~0ull; }
            act = (p < at ? at - a : b - at) >= 2u;
        } else { S.segA[nxt][p] = p; S.segB[nxt][p] = p + 1u; }
    }
    // the level's nodes are appended to their kind's list: one atomic per wave and kind
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (uint32_t k = 0; k < 2u; k++) {
        const unsigned long long m = __ballot(new_id != SW_NONE && kind == k);
        if (m == 0ull) continue;
        uint32_t base = 0;
        if (lane == (uint32_t)(__ffsll((long long)m) - 1)) base = atomicAdd(&S.list_count[k], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)m) - 1);
        if (new_id != SW_NONE && kind == k) S.list[k][base + (uint32_t)__popcll(m & lt)] = new_id;
    }
    const int any = __syncthreads_or(act ? 1 : 0);
    if (threadIdx.x == 0) S.tile_act[nxt][t] = (uint8_t)(any != 0);
}

// per tile and axis: left-going elements of the segment that is still open at the tile's end
__global__ __launch_bounds__(SW_TILE) void k_sw_pagg(SwState S, int cur) {
    __shared__ uint32_t s_cnt[SW_TILE / 64];
    const uint32_t t = blockIdx.x, axis = blockIdx.y, t0 = t * SW_TILE, t1 = min(t0 + (uint32_t)SW_TILE, S.N), p = t0 + threadIdx.x;
    if (!S.tile_act[cur][t]) return;
    const uint32_t a_last = S.segA[cur][t1 - 1u], b_last = S.segB[cur][a_last];
    if (b_last <= t1) return;
    const bool lf = p < t1 && p >= a_last && !S.right_side[S.ord[cur][axis][p]];
    const unsigned long long m = __ballot(lf);
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t c = 0; for (int w = 0; w < SW_TILE / 64; w++) c += s_cnt[w]; S.cf[axis][t] = c; }
}

// stable partition of every cut segment in one axis order; along axis 0 the elements that land in a segment of one position are their node's leaf children
__global__ __launch_bounds__(SW_TILE) void k_sw_part(SwState S, int cur, BinTree tr) {
    __shared__ unsigned long long s_bal[SW_TILE / 64];
    __shared__ uint32_t s_red[SW_TILE / 64];
    const uint32_t t = blockIdx.x, axis = blockIdx.y, t0 = t * SW_TILE, t1 = min(t0 + (uint32_t)SW_TILE, S.N), p = t0 + threadIdx.x;
    const int nxt = cur ^ 1;
    if (!S.tile_act[cur][t]) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool valid = p < t1;
    uint32_t a = p, b = p + 1u, e = 0;
    bool lf = false;
    if (valid) { a = S.segA[cur][p]; b = S.segB[cur][a]; e = S.ord[cur][axis][p]; lf = b - a >= 2u && !S.right_side[e]; }
    // left-goers of the segment open at the tile's start in the tiles before
    const uint32_t a_first = S.segA[cur][t0];
    uint32_t carry = 0;
    if (a_first < t0 && S.segB[cur][a_first] - a_first >= 2u) {
        uint32_t c = sw_walk_count(S.cf[axis], S.cf2[axis], a_first / SW_TILE, t);
        for (int o = 32; o >= 1; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o);
        if (lane == 0u) s_red[wave] = c;
        __syncthreads();
        for (int w = 0; w < SW_TILE / 64; w++) carry += s_red[w];
    }
    const unsigned long long m = __ballot(lf);
    if (lane == 0u) s_bal[wave] = m;
    __syncthreads();
    if (!valid) return;
    if (b - a < 2u) { S.ord[nxt][axis][p] = e; return; }
    // left-goers among [max(a, t0), p)
    const uint32_t lo = a > t0 ? a : t0;
    uint32_t nl = a < t0 ? carry : 0u;
    for (uint32_t w = 0; w <= wave; w++) {
        const uint32_t w0 = t0 + 64u * w;                                    // positions w0 .. w0 + 63
        const uint32_t from = lo > w0 ? lo - w0 : 0u, to = p > w0 ? (p - w0 < 64u ? p - w0 : 64u) : 0u;   // bits [from, to)
        if (to > from) {
            const unsigned long long hi_mask = to >= 64u ? ~0ull : ((1ull << to) - 1ull), lo_mask = (1ull << from) - 1ull;
            nl += (uint32_t)__popcll(s_bal[w] & hi_mask & ~lo_mask);
        }
    }
    int best_axis; uint32_t at;
    split_decode(S.best[cur][a], a, b, best_axis, at);
    const uint32_t np = lf ? a + nl : at + (p - a - nl);
    S.ord[nxt][axis][np] = e;
    if (axis == 0u) {
        if (lf && at - a == 1u) tr.left[sw_node_id(S, a, b, S.segR[cur][a])] = S.eref[e];
        if (!lf && b - at == 1u) tr.right[sw_node_id(S, a, b, S.segR[cur][a])] = S.eref[e];
    }
}

// boxes and collapse tables of the nodes of one level, children first (levels are processed deepest first; the top trees after the cluster trees they stand on)
__global__ void k_sw_up(const uint32_t* list, uint32_t n, BinTree tr, const Box* leaf_boxes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = list[i], l = tr.left[id], r = tr.right[id];
    float cl[7], cr[7];
    Box bl, br;
    if (l & REF_LEAF) { bl = canon_box(leaf_boxes[l & ~REF_LEAF]); for (int k = 0; k < 7; k++) cl[k] = 0.0f; } else { bl = tr.box[l]; for (int k = 0; k < 7; k++) cl[k] = tr.cost[7 * (size_t)l + k]; }
    if (r & REF_LEAF) { br = canon_box(leaf_boxes[r & ~REF_LEAF]); for (int k = 0; k < 7; k++) cr[k] = 0.0f; } else { br = tr.box[r]; for (int k = 0; k < 7; k++) cr[k] = tr.cost[7 * (size_t)r + k]; }
    Box b = empty_box(); grow_box(b, bl); grow_box(b, br);
    float cn[7]; uint8_t sp[8];
    collapse_table(cl, cr, box_area(b), cn, sp);
    tr.box[id] = b;
    for (int k = 0; k < 7; k++) tr.cost[7 * (size_t)id + k] = cn[k];
    for (int k = 0; k < 8; k++) tr.split[8 * (size_t)id + k] = sp[k];
}

// ---- which cluster owns which primitive and which PLOC node ----
__global__ void k_sw_roots(const uint32_t* cref, uint32_t c, uint32_t* leaf_cluster, uint32_t* node_cluster) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const uint32_t r = cref[i];
    if (r & REF_LEAF) leaf_cluster[r & ~REF_LEAF] = i; else node_cluster[r] = i;
}
__global__ void k_sw_parents(BinTree tr, uint32_t node_base, uint32_t* parent_leaf, uint32_t* parent_node) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= node_base) return;
    const uint32_t l = tr.left[id], r = tr.right[id];
    if (l & REF_LEAF) parent_leaf[l & ~REF_LEAF] = id; else parent_node[l] = id;
    if (r & REF_LEAF) parent_leaf[r & ~REF_LEAF] = id; else parent_node[r] = id;
}
__global__ void k_sw_owner(uint32_t n, uint32_t node_base, const uint32_t* parent_leaf, const uint32_t* parent_node, uint32_t* leaf_cluster, const uint32_t* node_cluster, uint32_t* node_owner) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (leaf_cluster[i] == SW_NONE) { uint32_t cur = parent_leaf[i]; while (node_cluster[cur] == SW_NONE) cur = parent_node[cur]; leaf_cluster[i] = node_cluster[cur]; }
    } else if (i - n < node_base) {
        uint32_t cur = i - n; while (node_cluster[cur] == SW_NONE) cur = parent_node[cur];
        node_owner[i - n] = node_cluster[cur];
    }
}
__global__ void k_sw_iota(uint32_t n, uint32_t* v) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = i; }

// first position of every group: the clusters' primitive counts are scanned in place (k_sw_cluster_sizes, then an exclusive scan over ngc + 1 entries), the trees'
// first clusters are found where the segment number changes (k_sw_tree_first, after the scan)
__global__ void k_sw_cluster_sizes(const uint32_t* cref, uint32_t c, BinTree tr, uint32_t* gsize) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) { const uint32_t r = cref[i]; gsize[i] = (r & REF_LEAF) ? 1u : tr.count[r]; }
    if (i == c) gsize[c] = 0u;
}
__global__ void k_sw_tree_first(const uint32_t* cseg, uint32_t c, uint32_t ngc, uint32_t nseg, uint32_t P0, uint32_t* grp_first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c && (i == 0u || (cseg && cseg[i - 1u] != cseg[i]))) grp_first[ngc + (cseg ? cseg[i] : 0u)] = P0 + i;
    if (i == c) grp_first[ngc + nseg] = P0 + c;
}
// elements in base order: primitives grouped by cluster (prim_sorted / prim_group: the by-cluster sort's output), then the clusters
__global__ void k_sw_elements(uint32_t N, uint32_t P0, uint32_t ngc, const uint32_t* prim_sorted, const uint32_t* prim_group, const Box* prim_boxes,
                              const uint32_t* cref, const Box* cbox, const uint32_t* cseg, BinTree tr, Box* ebox, uint32_t* ecnt, uint32_t* eref, uint32_t* egrp) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    if (e < P0) { const uint32_t pr = prim_sorted[e]; ebox[e] = canon_box(prim_boxes[pr]); ecnt[e] = 1u; eref[e] = REF_LEAF | pr; egrp[e] = prim_group[e]; }
    else { const uint32_t i = e - P0, r = cref[i]; ebox[e] = canon_box(cbox[i]); ecnt[e] = (r & REF_LEAF) ? 1u : tr.count[r]; eref[e] = r; egrp[e] = ngc + (cseg ? cseg[i] : 0u); }
}
__global__ void k_sw_keys(uint32_t N, const Box* ebox, int axis, uint32_t* keys, uint32_t* vals) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < N) { keys[e] = sweep_key(ebox[e], axis); vals[e] = e; }
}
// root of every tree and its box
__global__ void k_sw_tree_roots(SwState S, const uint32_t* grp_first, uint32_t nseg, BinTree tr, uint32_t* root_ref, Box* root_box) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nseg) return;
    const uint32_t a = grp_first[S.ngc + j], b = grp_first[S.ngc + j + 1u];
    if (b - a == 1u) { root_ref[j] = S.eref[a]; root_box[j] = S.ebox[a]; }      // a tree of one cluster: base position = element
    else { const uint32_t id = S.node_base + (b - S.P0) - j - 2u; root_ref[j] = id; root_box[j] = tr.box[id]; }
}

}  // namespace msne
