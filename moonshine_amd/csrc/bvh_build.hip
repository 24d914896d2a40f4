// bvh_build.hip — GPU BVH build: replaces the driver's vkCmdBuildAccelerationStructuresKHR for BLAS and
// TLAS (engine/hrtsystem/Accel.zig:94-184 makeBlases, :484 TLAS build, :629-679 recordRebuild).
//
// Pipeline (all HIP kernels, host only sequences launches and reads round / level counts):
//   k_prim_boxes / k_bounds → k_morton (63-bit: 21 bits per axis) → stable LSD radix sort, 4 x 8 bit on the low word then 4 x 8 bit on the high word (k_radix_hist / _scan_digit / _scatter)
//   → PLOC rounds (k_ploc_nn / _mark / _scan / _merge: bottom-up agglomerative clustering, boxes come with the merges) down to the cluster count top_clusters() asks for
//     (none at all up to 32 M primitives)
//   → the top-down surface-area sweep over what is left (bvh_sweep.h: k_sw_*; one tree level per pass, every tree of the batch at once)
//   → k_collapse (level-synchronous collapse to 8-wide in octant slot order, one item per leaf, quantised 80-B nodes)
//   → k_emit_tris / k_emit_items; TLAS leaf records (k_tlas_leaves); in-place TLAS re-fit for transform edits (k_tlas_refit_*).
// The same builder serves BLAS (items = triangles) and TLAS (items = instances).
#include "msne_device.h"
#include "../host/host.h"
#include "bvh_topdown.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <atomic>
#include <chrono>
#include <exception>

namespace msne {


// ---------------- helpers ----------------
__device__ __forceinline__ uint32_t float_to_ordered(float f) { uint32_t u = f2u(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__host__ __device__ __forceinline__ float ordered_to_float(uint32_t u) { return u2f((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }


// ---------------- primitive boxes ----------------
// triangles of one BLAS: geometry g -> mesh; prim boxes + source records
// `geo` = geometry index inside its instance; `inst` = owning instance for triangles of the merged world BLAS (else unused)
// normals / texcoords: the mesh's attribute arrays (attr_count elements; an index beyond them reads as zero — the render refuses such a
// mesh before any shading) or nullptr; `indexed`: attributes are read by vertex index (glTF) instead of by corner 3 * prim + k (Hydra)
struct BlasGeo { const float* positions; const uint32_t* indices; uint32_t tri_offset, tri_count, geo, inst; const float* normals; const float* texcoords; uint32_t indexed, attr_count; };

// geometry that owns triangle i of the concatenated list (tri_offset ascending): binary search — a world BLAS can merge 10^5 geometries
__device__ __forceinline__ uint32_t geo_of(const BlasGeo* geos, uint32_t ngeo, uint32_t i) {
    uint32_t lo = 0, hi = ngeo;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (geos[mid].tri_offset <= i) lo = mid; else hi = mid; }
    return lo;
}

__global__ void k_prim_boxes_tris(const BlasGeo* geos, uint32_t ngeo, uint32_t n, Box* boxes) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BlasGeo ge = geos[geo_of(geos, ngeo, i)];
    const uint32_t p = i - ge.tri_offset;
    const uint32_t i0 = ge.indices[3 * p], i1 = ge.indices[3 * p + 1], i2 = ge.indices[3 * p + 2];
    Box b;
    for (int k = 0; k < 3; k++) {
        float a = ge.positions[3 * (size_t)i0 + k], bb = ge.positions[3 * (size_t)i1 + k], c = ge.positions[3 * (size_t)i2 + k];
        // a triangle with a corner that is not a number cannot be hit (the watertight test accepts nothing that is not a number): min / max skip the NaN and the box is
        // that of the other corners.  An INFINITE corner is the same triangle to the test and is read the same way here — kept, it made every box above it infinite
        // (one such vertex cost a scene a quarter of its hits: tests/test_gpu_parity.py::test_triangles_with_a_vertex_that_is_not_finite)
        const float nan = u2f(0x7fc00000u);
        a = absf(a) < 3.4e38f ? a : nan; bb = absf(bb) < 3.4e38f ? bb : nan; c = absf(c) < 3.4e38f ? c : nan;
        b.lo[k] = fminf(a, fminf(bb, c)); b.hi[k] = fmaxf(a, fmaxf(bb, c));
    }
    boxes[i] = b;
}

__global__ void k_bounds(const Box* boxes, uint32_t n, uint32_t* bounds /*6 ordered uints: lo xyz, hi xyz*/) {
    __shared__ uint32_t s[6];
    if (threadIdx.x < 6) s[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Box b = boxes[i];
        for (int k = 0; k < 3; k++) { atomicMin(&s[k], float_to_ordered(b.lo[k])); atomicMax(&s[3 + k], float_to_ordered(b.hi[k])); }
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s[threadIdx.x]);
}

// ---- batched builds: several independent trees ("segments": the BLASes of one rebuild) go through ONE pass of the builder ----
// Primitives of a segment are consecutive in the input; seg_first[j] is the first primitive of segment j (seg_first[nseg] = n).
// Every segment gets its own Morton frame, the sort keeps segments apart, PLOC only pairs clusters of one segment and stops at one
// cluster per segment: each tree is exactly the tree a build of that segment alone produces.
__global__ void k_seg_of(const uint32_t* seg_first, uint32_t nseg, uint32_t n, uint32_t* seg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg_first[mid] <= i) lo = mid; else hi = mid; }
    seg[i] = lo;
}
__global__ void k_bounds_seg(const Box* boxes, const uint32_t* seg, uint32_t n, uint32_t* bounds /*6 ordered uints per segment*/) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const uint32_t sg = valid ? seg[i] : 0xFFFFFFFFu;
    uint32_t v[6];
    if (valid) { const Box b = boxes[i]; for (int k = 0; k < 3; k++) { v[k] = float_to_ordered(b.lo[k]); v[3 + k] = float_to_ordered(b.hi[k]); } }
    else for (int k = 0; k < 3; k++) { v[k] = 0xFFFFFFFFu; v[3 + k] = 0u; }
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)sg);
    if (__ballot(valid && sg != first) == 0ull && __ballot(!valid && first != 0xFFFFFFFFu) == 0ull) {   // the whole wave lies in one segment: reduce, then six atomics
        for (int o = 32; o >= 1; o >>= 1) for (int k = 0; k < 3; k++) { v[k] = min(v[k], (uint32_t)__shfl_xor((int)v[k], o)); v[3 + k] = max(v[3 + k], (uint32_t)__shfl_xor((int)v[3 + k], o)); }
        if ((threadIdx.x & 63u) == 0 && valid) for (int k = 0; k < 3; k++) { atomicMin(&bounds[6 * (size_t)sg + k], v[k]); atomicMax(&bounds[6 * (size_t)sg + 3 + k], v[3 + k]); }
    } else if (valid) for (int k = 0; k < 3; k++) { atomicMin(&bounds[6 * (size_t)sg + k], v[k]); atomicMax(&bounds[6 * (size_t)sg + 3 + k], v[3 + k]); }
}
__global__ void k_fill_bounds(uint32_t* bounds, uint32_t nseg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * nseg) bounds[i] = (i % 6) < 3 ? 0xFFFFFFFFu : 0u;
}
__global__ void k_gather_u32(const uint32_t* src, const uint32_t* idx, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// Morton codes of up to 63 bits: `cells` = 2^10 or 2^21 per axis of the primitive's centre inside its segment's bounds (a 30-bit code puts ~60 primitives of a
// 64 M-triangle scene in one cell — more than PLOC's search radius — and the sort then leaves them in input order).  Low word -> keys, high word (31 bits) -> keys_hi.
__device__ __forceinline__ unsigned long long expand_bits21(unsigned long long x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__global__ void k_morton(const Box* boxes, uint32_t n, const uint32_t* bounds_all, const uint32_t* seg /* nullptr: one segment */, float cells /* per axis: 2^10 or 2^21 */, uint32_t* keys, uint32_t* keys_hi, uint32_t* idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Box b = boxes[i];
    const uint32_t* bounds = bounds_all + (seg ? 6 * (size_t)seg[i] : 0);
    unsigned long long code = 0;
    for (int k = 0; k < 3; k++) {
        const float lo = ordered_to_float(bounds[k]), hi = ordered_to_float(bounds[3 + k]);
        const float ext = hi - lo;
        const float c = (b.lo[k] + b.hi[k]) * 0.5f;
        float x = ext > 0.0f ? (c - lo) / ext : 0.0f;
        x = fminf(fmaxf(x * cells, 0.0f), cells - 1.0f);
        code |= expand_bits21((unsigned long long)(uint32_t)x) << (2 - k);
    }
    keys[i] = (uint32_t)code; keys_hi[i] = (uint32_t)(code >> 32); idx[i] = i;
}

// exclusive prefix sum over the 1024 threads of a workgroup (wave shuffles + one LDS hop); returns the thread's exclusive prefix, `total` = sum of all
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t* s_wave /* [16] */, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += t; }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < 16u ? s_wave[lane] : 0u, wi = w;
        for (int o = 1; o < 16; o <<= 1) { const uint32_t t = __shfl_up(wi, o); if (lane >= (uint32_t)o) wi += t; }
        if (lane < 16u) s_wave[lane] = wi - w;       // exclusive prefix of the wave totals
        if (lane == 15u) s_wave[16] = wi;            // grand total
    }
    __syncthreads();
    total = s_wave[16];
    return s_wave[wave] + inc - v;
}

// ---------------- LSD radix sort (stable), 8 bits per pass ----------------
constexpr int RS_TILE = 2048;
__global__ __launch_bounds__(256) void k_radix_hist(const uint32_t* keys, uint32_t n, int shift, uint32_t* ghist, uint32_t ntiles) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    for (uint32_t j = threadIdx.x; j < RS_TILE; j += 256) { uint32_t i = base + j; if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u); }
    __syncthreads();
    ghist[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of `total` counters in place by ONE workgroup (small inputs: PLOC block sums, group sizes)
__global__ __launch_bounds__(1024) void k_radix_scan(uint32_t* ghist, uint32_t total) {
    __shared__ uint32_t s_wave[17];
    const uint32_t per = (total + 1023u) / 1024u;
    const uint32_t a = threadIdx.x * per < total ? threadIdx.x * per : total, b = a + per < total ? a + per : total;
    uint32_t s = 0;
    for (uint32_t i = a; i < b; i++) s += ghist[i];
    uint32_t all;
    uint32_t run = block_scan_1024(s, s_wave, all);
    for (uint32_t i = a; i < b; i++) { uint32_t v = ghist[i]; ghist[i] = run; run += v; }
}
// the sort's histogram (digit-major: ghist[digit * ntiles + tile]) scanned by one workgroup PER DIGIT: exclusive prefix over the digit's tiles in place, the digit's
// total to digit_total[digit]; k_radix_scatter adds the totals of the smaller digits itself (a single workgroup over all 256 x ntiles counters took 190 us per pass
// for a million keys — more than histogram and scatter together)
__global__ __launch_bounds__(256) void k_radix_scan_digit(uint32_t* ghist, uint32_t ntiles, uint32_t* digit_total) {
    __shared__ uint32_t s_w[4];
    uint32_t* h = ghist + (size_t)blockIdx.x * ntiles;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < ntiles; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < ntiles ? h[i] : 0u;
        uint32_t inc = v;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (lane >= (uint32_t)o) inc += t; }
        __syncthreads();
        if (lane == 63u) s_w[wave] = inc;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 4u; w++) { if (w < wave) before += s_w[w]; all += s_w[w]; }
        if (i < ntiles) h[i] = carry + before + inc - v;
        carry += all;
    }
    if (threadIdx.x == 0) digit_total[blockIdx.x] = carry;
}
// one wave per tile, elements scattered in order → stable
__global__ __launch_bounds__(64) void k_radix_scatter(const uint32_t* keys, const uint32_t* vals, uint32_t n, int shift,
                                                       const uint32_t* ghist, const uint32_t* digit_total, uint32_t ntiles, uint32_t* okeys, uint32_t* ovals) {
    __shared__ uint32_t base[256];
    {   // first output position of every digit in this tile: keys with smaller digits anywhere + keys with this digit in the tiles before
        uint32_t tot[4], run = 0;
        for (int q = 0; q < 4; q++) { tot[q] = digit_total[threadIdx.x * 4 + q]; run += tot[q]; }
        uint32_t inc = run;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if (threadIdx.x >= (uint32_t)o) inc += t; }
        uint32_t ex = inc - run;
        for (int q = 0; q < 4; q++) { const int j = threadIdx.x * 4 + q; base[j] = ex + ghist[(size_t)j * ntiles + blockIdx.x]; ex += tot[q]; }
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t tbase = blockIdx.x * RS_TILE;
    for (uint32_t c = 0; c < RS_TILE; c += 64) {
        const uint32_t i = tbase + c + lane;
        const bool valid = i < n;
        const uint32_t k = valid ? keys[i] : 0u, v = valid ? vals[i] : 0u;
        const uint32_t d = (k >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bm = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bm : ~bm;
        }
        const uint32_t rank = __popcll(peers & lt);
        uint32_t pos = 0;
        if (valid) pos = base[d] + rank;
        __syncthreads();
        if (valid && rank == 0) base[d] += (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) { okeys[pos] = k; ovals[pos] = v; }
    }
}

// ---------------- binary tree (built bottom-up by PLOC, then collapsed to 8-wide) ----------------
constexpr uint32_t REF_LEAF = 0x80000000u;
struct BinTree {
    uint32_t* left; uint32_t* right;         // child refs of internal node i (bit 31 = leaf: index into the sorted primitives)
    Box* box;                                // internal node boxes
    float* cost;                             // [7 per node] cost[k-1] = cheapest way to put the subtree into k slots of a wide node (k = 1..7)
    uint8_t* split;                          // [8 per node] split[j-1], j = 2..8: slots given to the left child when the subtree is spread over j slots
                                             //              (0: k slots are not worth it, use j - 1); split[0] unused
    uint32_t* count;                         // primitives below the node (the weights of the top-down stage)
};
// (the tables' recurrence — the optimal collapse into 8-wide nodes — is collapse_table() in bvh_topdown.h, shared with the host's top-down stages)

__global__ void k_gather_boxes(const Box* boxes, const uint32_t* idx, uint32_t n, Box* sorted) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) sorted[i] = boxes[idx[i]];
}

// ---------------- PLOC: parallel locally-ordered clustering (Meister & Bittner 2018) ----------------
// Bottom-up agglomeration over the Morton-sorted primitives: every cluster looks PLOC_RADIUS neighbours to either side
// for the partner with the smallest merged surface area; mutual choices merge.  Candidate pairs are ordered by
// (area, min index, max index), a key both ends agree on, so the globally best pair is always mutual and every round
// makes progress.  Node ids and output slots come from prefix sums (no atomics): the tree is deterministic.
// The radius: 16 looked best when PLOC went in (round 1), but a sweep over the bench scenes says 4 — S2 3298 Mrays/s against 3128 at 16 (3225 at 3, 3210 at 6), S1 and
// the 16 M-triangle grid +0.3 … +0.9 %, the stand-in unchanged (profiles/r03_ploc_radius.txt).  A wider search finds a smaller merged box for the pair at hand but
// leaves stragglers that join late and high up in the tree.  Picking among candidate radii by the tree's surface-area cost was tried and does not pay: the cost ranks
// the trees roughly (tools/bvh_sah.py) but not finely — the cheapest TLAS by area (radius 3) traced 3.4 % slower than the radius-4 one.
constexpr int PLOC_RADIUS = 4;
constexpr int PLOC_BLOCK = 256;

// Round state lives on the device so that the host can queue several rounds without reading anything back: round r reads slot r & 1
// and writes the other one.  c = clusters left, node_base = binary nodes made so far, stuck = a round merged nothing (cannot happen).
struct PlocState { uint32_t c, node_base, stuck, pad; };

__global__ __launch_bounds__(PLOC_BLOCK) void k_ploc_nn(const Box* cbox, const uint32_t* cseg /* nullptr: one segment */, const PlocState* st, uint32_t radius, uint32_t* nn) {
    const uint32_t c = st->c;
    const uint32_t i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    if (i >= c) return;
    const Box bi = cbox[i];
    const uint32_t lo = i > radius ? i - radius : 0u, hi = i + radius < c ? i + radius : c - 1u;
    const uint32_t si = cseg ? cseg[i] : 0u;
    float best = 3.0e38f; uint32_t bj = i;
    for (uint32_t j = lo; j <= hi; j++) {
        if (j == i) continue;
        if (cseg && cseg[j] != si) continue;   // clusters of other trees are not neighbours
        const Box bb = cbox[j];
        Box u;
        for (int k = 0; k < 3; k++) { u.lo[k] = fminf(bi.lo[k], bb.lo[k]); u.hi[k] = fmaxf(bi.hi[k], bb.hi[k]); }
        const float a = box_area(u);
        // same area: the pair with the smaller (min, max) index wins — for a fixed i that is simply the smaller j
        if (a < best || (a == best && ((j < i) != (bj < i) ? j < i : j < bj))) { best = a; bj = j; }
    }
    nn[i] = bj;
}

// flags per cluster: bit 0 = survives into the next round (alone, or as the owner of a merge), bit 16 = owner of a merge
__global__ __launch_bounds__(PLOC_BLOCK) void k_ploc_mark(const uint32_t* nn, const PlocState* st, uint32_t* flags, uint32_t* block_sums) {
    __shared__ uint32_t wsum[PLOC_BLOCK / 64];
    const uint32_t c = st->c;
    if (blockIdx.x * PLOC_BLOCK >= c) return;     // (the grid is sized for the cluster count of an earlier round)
    const uint32_t i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    uint32_t f = 0;
    if (i < c) {
        const uint32_t j = nn[i];
        const bool mutual = j != i && nn[j] == i;
        f = mutual ? (i < j ? 0x10001u : 0u) : 1u;
        flags[i] = f;
    }
    uint32_t s = f;
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < PLOC_BLOCK / 64; w++) t += wsum[w]; block_sums[blockIdx.x] = t; }   // < 2^16 per half: no carry
}

// exclusive scan of the per-block sums (both 16-bit halves at once would overflow: they are widened to two words here)
__global__ __launch_bounds__(1024) void k_ploc_scan(const uint32_t* block_sums, const PlocState* st, PlocState* st_next, uint2* block_base, uint32_t target /* clusters at the end: one per segment */) {
    __shared__ uint32_t s_wave[17];
    const uint32_t nblocks = (st->c + PLOC_BLOCK - 1) / PLOC_BLOCK;
    const uint32_t per = (nblocks + 1023u) / 1024u;
    const uint32_t a = threadIdx.x * per < nblocks ? threadIdx.x * per : nblocks, b = a + per < nblocks ? a + per : nblocks;
    uint2 s = make_uint2(0, 0);
    for (uint32_t i = a; i < b; i++) { const uint32_t v = block_sums[i]; s.x += v & 0xffffu; s.y += v >> 16; }
    uint2 run, tot;
    run.x = block_scan_1024(s.x, s_wave, tot.x);
    __syncthreads();      // (s_wave is reused by the second scan)
    run.y = block_scan_1024(s.y, s_wave, tot.y);
    if (threadIdx.x == 0) {
        st_next->c = tot.x; st_next->node_base = st->node_base + tot.y;
        st_next->stuck = st->stuck | ((st->c > target && tot.y == 0u) ? 1u : 0u);
    }
    for (uint32_t i = a; i < b; i++) { const uint32_t v = block_sums[i]; block_base[i] = run; run.x += v & 0xffffu; run.y += v >> 16; }
}

__global__ __launch_bounds__(PLOC_BLOCK) void k_ploc_merge(const uint32_t* cref, const Box* cbox, const uint32_t* cseg, const uint32_t* nn, const uint32_t* flags, const uint2* block_base,
                                                        const PlocState* st, BinTree t, uint32_t* oref, Box* obox, uint32_t* oseg) {
    __shared__ uint2 wbase[PLOC_BLOCK / 64];
    const uint32_t c = st->c, node_base = st->node_base;
    if (blockIdx.x * PLOC_BLOCK >= c) return;
    const uint32_t i = blockIdx.x * PLOC_BLOCK + threadIdx.x;
    const uint32_t f = i < c ? flags[i] : 0u;
    const unsigned long long keep = __ballot(f & 1u), own = __ballot(f >> 16);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) wbase[threadIdx.x >> 6] = make_uint2((uint32_t)__popcll(keep), (uint32_t)__popcll(own));
    __syncthreads();
    uint2 base = block_base[blockIdx.x];
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) { base.x += wbase[w].x; base.y += wbase[w].y; }
    if (!(f & 1u)) return;
    const uint32_t pos = base.x + (uint32_t)__popcll(keep & lt);
    uint32_t ref = cref[i]; Box b = cbox[i];
    if (f >> 16) {
        const uint32_t j = nn[i], id = node_base + base.y + (uint32_t)__popcll(own & lt);
        const uint32_t rj = cref[j]; const Box bj = cbox[j];
        for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], bj.lo[k]); b.hi[k] = fmaxf(b.hi[k], bj.hi[k]); }
        t.left[id] = ref; t.right[id] = rj; t.box[id] = b;
        {   // children are final (made in earlier rounds): fill this node's table now
            float cl[7], cr[7];
            if (ref & REF_LEAF) { for (int k = 0; k < 7; k++) cl[k] = 0.0f; } else for (int k = 0; k < 7; k++) cl[k] = t.cost[7 * (size_t)ref + k];
            if (rj & REF_LEAF) { for (int k = 0; k < 7; k++) cr[k] = 0.0f; } else for (int k = 0; k < 7; k++) cr[k] = t.cost[7 * (size_t)rj + k];
            float cn[7]; uint8_t sp[8];
            collapse_table(cl, cr, box_area(b), cn, sp);
            for (int k = 0; k < 7; k++) t.cost[7 * (size_t)id + k] = cn[k];
            for (int k = 0; k < 8; k++) t.split[8 * (size_t)id + k] = sp[k];
            t.count[id] = ((ref & REF_LEAF) ? 1u : t.count[ref]) + ((rj & REF_LEAF) ? 1u : t.count[rj]);
        }
        ref = id;
    }
    oref[pos] = ref; obox[pos] = b;
    if (cseg) oseg[pos] = cseg[i];
}

__global__ void k_ploc_init(uint32_t n, uint32_t* cref) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cref[i] = REF_LEAF | i;
}

// ---------------- collapse to 8-wide + quantise ----------------
struct CollapseWork { uint32_t bin; uint32_t wide; };

// A node's grid (Node8: origin + one power-of-two quantum per axis, children as bytes on it).  The quantum: the smallest power of two s with ext / s <= 252 — three
// steps of headroom: one for the outward rounding of a child's upper face, one for the origin below, one for the rounding of that origin.  The ORIGIN lies one quantum
// BELOW the node's lower face: a child on that face then quantises to floor(1 - 1e-3) = 0 and its plane sits a quantum outside it, like every other plane of the grid
// (each is pushed outward by at least 1e-3 quantum).  Before, such a child's plane was the face itself, to the bit — and a ray that starts within the underflow range
// of that plane (1e-45 beside it, or on it and leaving by a direction component of -1e-45) was out of the box at t = 0 while the triangle test, whose products underflow,
// still took the edge: the one place where a box was NOT wider than its triangles by a margin the arithmetic cannot cross (tests/test_gpu_parity.py::test_lattice_rays).
__device__ __forceinline__ uint8_t grid_exponent(float lo, float hi) {
    const float ext = hi - lo;
    int ex = 1;
    if (ext > 0.0f) { const float q = ext / 252.0f; ex = (int)((f2u(q) >> 23) & 0xff) + 1; }
    if (ex < 1) ex = 1;
    if (ex > 254) ex = 254;
    return (uint8_t)ex;
}
// No axis of a grid finer than a quarter of its coarsest: a node of coplanar children (extent 0 in one axis: quantum 2^-126) otherwise has NO margin in that axis, and the
// triangle test's t is only good to ~1e-7 of the triangle's size — a ray that starts in such a triangle's plane is given t = +2e-8 by it and was already out of the box
// ("The oracle, corrected by the product").  With this the flat axis keeps 1e-3 quantum = ~1e-6 of the node's largest extent.
__device__ __forceinline__ void grid_no_axis_much_finer(uint8_t e[3]) {
    const int em = max((int)e[0], max((int)e[1], (int)e[2]));
    for (int k = 0; k < 3; k++) if ((int)e[k] < em - 2) e[k] = (uint8_t)(em - 2);
}
// Every child box is grown by 1e-4 of its own largest extent, on every side, before it is quantised.  What the watertight test's t is worth depends on the TRIANGLE:
// its weights are differences of products, and a needle (15.8 x 0.028) or a ray 1.4 degrees off the plane puts the computed hit point eps x size x (aspect ratio, or
// 1 / sin) away from the true one ALONG THE RAY — in round 6's sweeps +3.7e-4 / +6.8e-4 for true distances of -1.5e-4 / -2.4e-3, hits the search over every triangle takes
// (the test oracle's box test carries the same term; tests/test_oracle.py::test_hull_films_with_and_without_boxes).  No bound exists in what a node knows, but the error is
// always a fraction of the triangle's size and a triangle is no larger than a box that holds it: 1e-4 covers amplifications up to ~1600.  Children of a child are grown by
// less than it is, so boxes stay nested; the grid's three quanta of headroom (>= 1.2e-2 of the extent) hold it.
#ifndef MSNE_BOX_GROWTH
#define MSNE_BOX_GROWTH 1e-4f   // (0: measurements only — tools/variant_rates.py)
#endif
__device__ __forceinline__ float box_growth(const Box& b) {
    const float g = fmaxf(fmaxf(b.hi[0] - b.lo[0], b.hi[1] - b.lo[1]), b.hi[2] - b.lo[2]);
    return (g > 0.0f && g < 3.0e38f) ? MSNE_BOX_GROWTH * g : 0.0f;   // (an empty or unbounded box: as it is)
}
__device__ __forceinline__ float grid_origin(float lo, uint8_t ex) {
    const float o = lo - u2f((uint32_t)ex << 23);
    return (o == o && o > -3.0e38f) ? o : lo;   // (a box at the end of the range or not a number: as it is)
}


__global__ void k_collapse(const CollapseWork* work, uint32_t nwork, CollapseWork* next, uint32_t* next_count,
                           BinTree t, const Box* leaf_boxes, const uint32_t* sorted_idx,
                           Node8* nodes, uint32_t* node_counter, uint32_t* item_counter, uint32_t* item_src) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwork) return;
    const uint32_t bin = work[w].bin, widx = work[w].wide;
    // Every leaf holds ONE primitive: a watertight triangle test costs about five quantised box tests in k_trace_*, so it
    // pays to box every triangle on its own (S1: 9.3 -> 4.2 triangle tests per ray for 12.5 -> 13.6 node visits).  Which
    // binary nodes become wide nodes is the optimal cut of BinTree::cost (S1: 4.1 -> 6.3 children per node against opening
    // the largest child until the node is full).
    uint32_t ch[8]; int nch = 0;
    auto ref_box = [&](uint32_t r) -> Box { return (r & REF_LEAF) ? leaf_boxes[r & ~REF_LEAF] : t.box[r]; };
    if (bin & REF_LEAF) ch[nch++] = bin;
    else { ch[nch++] = t.left[bin]; ch[nch++] = t.right[bin]; }
    // the optimal cut below `bin` (see BinTree): spread its 8 slots over the two children as the table says
    if (!(bin & REF_LEAF)) {
        nch = 0;
        uint32_t sn[8]; uint8_t sj[8]; int sp = 0;
        { const uint32_t k = t.split[8 * (size_t)bin + 7]; sn[sp] = t.right[bin]; sj[sp++] = (uint8_t)(8 - k); sn[sp] = t.left[bin]; sj[sp++] = (uint8_t)k; }
        while (sp) {
            const uint32_t n_ = sn[--sp]; uint32_t j = sj[sp];
            if (n_ & REF_LEAF) { ch[nch++] = n_; continue; }
            while (j > 1 && t.split[8 * (size_t)n_ + j - 1] == 0) j--;   // fewer slots are as good
            if (j == 1) { ch[nch++] = n_; continue; }                    // stays one child: a wide node of its own
            const uint32_t k = t.split[8 * (size_t)n_ + j - 1];
            sn[sp] = t.right[n_]; sj[sp++] = (uint8_t)(j - k); sn[sp] = t.left[n_]; sj[sp++] = (uint8_t)k;
        }
    }
    Box cb[8]; Box nb;
    for (int k = 0; k < 3; k++) { nb.lo[k] = 3.0e38f; nb.hi[k] = -3.0e38f; }
    uint32_t n_internal = 0, n_items = 0;
    for (int i = 0; i < nch; i++) {
        cb[i] = ref_box(ch[i]);
        for (int k = 0; k < 3; k++) { nb.lo[k] = fminf(nb.lo[k], cb[i].lo[k]); nb.hi[k] = fmaxf(nb.hi[k], cb[i].hi[k]); }
        if (ch[i] & REF_LEAF) n_items++; else n_internal++;
    }
    // Octant placement: slot s stands for the corner direction (s&1 ? +x : -x, s&2 ? +y : -y, s&4 ? +z : -z) of the node.
    // Children are matched to slots greedily by the projection of (child centre - node centre) on that direction, so
    // the traversal kernel can order a node's hit children for a ray by (slot ^ ray octant) — a table lookup.
    int child_of_slot[8];
    {
        float proj[8][3];
        // (a child whose box is NaN — a triangle of NaN vertices — or a node of nothing else has no direction: projection 0.  Every round places ONE child that is
        // still waiting into ONE free slot whatever the numbers are: a child left out would leave its reserved work entry unwritten)
        for (int i = 0; i < nch; i++) for (int k = 0; k < 3; k++) { const float v = (cb[i].lo[k] + cb[i].hi[k]) - (nb.lo[k] + nb.hi[k]); proj[i][k] = (v == v && absf(v) < 3.0e38f) ? v : 0.0f; }
        for (int s = 0; s < 8; s++) child_of_slot[s] = -1;
        uint32_t child_done = 0;
        for (int r = 0; r < nch; r++) {
            float bc = 0.0f; int bi = -1, bs = -1;
            for (int i = 0; i < nch; i++) if (!((child_done >> i) & 1u)) for (int s = 0; s < 8; s++) if (child_of_slot[s] < 0) {
                const float c = ((s & 1) ? proj[i][0] : -proj[i][0]) + ((s & 2) ? proj[i][1] : -proj[i][1]) + ((s & 4) ? proj[i][2] : -proj[i][2]);
                if (bi < 0 || c > bc) { bc = c; bi = i; bs = s; }
            }
            child_of_slot[bs] = bi; child_done |= 1u << bi;
        }
    }
    const uint32_t child_base = n_internal ? atomicAdd(node_counter, n_internal) : 0u;
    const uint32_t item_base = n_items ? atomicAdd(item_counter, n_items) : 0u;
    const uint32_t qpos = n_internal ? atomicAdd(next_count, n_internal) : 0u;

    Node8 nd;
    uint8_t e[3]; float inv_s[3], grid_o[3];
    for (int k = 0; k < 3; k++) e[k] = grid_exponent(nb.lo[k], nb.hi[k]);
    grid_no_axis_much_finer(e);
    for (int k = 0; k < 3; k++) {
        inv_s[k] = u2f((uint32_t)(254 - e[k]) << 23);   // 2^-(ex-127)
        grid_o[k] = grid_origin(nb.lo[k], e[k]);
    }
    nd.ox = grid_o[0]; nd.oy = grid_o[1]; nd.oz = grid_o[2];
    nd.ex = e[0]; nd.ey = e[1]; nd.ez = e[2];
    uint32_t imask = 0, lmask = 0, ii = 0, io = 0;
    for (int i = 0; i < 8; i++) for (int k = 0; k < 3; k++) { nd.qlo[k][i] = 255; nd.qhi[k][i] = 0; }
    for (int i = 0; i < 7; i++) nd.pad[i] = 0;
    for (int s = 0; s < 8; s++) {   // internal children and leaf items are both numbered in slot order
        const int i = child_of_slot[s];
        if (i < 0) continue;
        for (int k = 0; k < 3; k++) {
            const float origin = k == 0 ? nd.ox : (k == 1 ? nd.oy : nd.oz);
            // outward by at least 1e-3 quantum: a ray that runs exactly in a face plane of the TRUE box (axis-parallel, reciprocal
            // 1e30) then sees the quantised plane at +-(1e-3 * scale * 1e30), far above the ~1e-5 * scale * 1e30 rounding noise of
            // q*a + b in the traversal; a child on the node's own lower face quantises to 0: one quantum below it (grid_origin)
            const float grow = box_growth(cb[i]);
            float ql = floorf((cb[i].lo[k] - grow - origin) * inv_s[k] - 1e-3f);
            float qh = ceilf((cb[i].hi[k] + grow - origin) * inv_s[k] + 1e-3f);
            ql = fminf(fmaxf(ql, 0.0f), 255.0f); qh = fminf(fmaxf(qh, 0.0f), 255.0f);
            nd.qlo[k][s] = (uint8_t)ql; nd.qhi[k][s] = (uint8_t)qh;
        }
        if (ch[i] & REF_LEAF) {
            lmask |= 1u << s;
            item_src[item_base + io++] = sorted_idx[ch[i] & ~REF_LEAF];
        } else {
            imask |= 1u << s;
            next[qpos + ii].bin = ch[i]; next[qpos + ii].wide = child_base + ii;
            ii++;
        }
    }
    nd.imask = (uint8_t)imask; nd.lmask = (uint8_t)lmask;
    nd.child_base = child_base; nd.item_base = item_base;
    nodes[widx] = nd;
}

__global__ void k_emit_tris(const BlasGeo* geos, uint32_t ngeo, const uint32_t* item_src, uint32_t item_begin, uint32_t n, TriRec* tris, TriRot* rots, TriAttr* attrs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = item_src[item_begin + i];
    const BlasGeo ge = geos[geo_of(geos, ngeo, src)];
    const uint32_t p = src - ge.tri_offset;
    const uint32_t i0 = ge.indices[3 * p], i1 = ge.indices[3 * p + 1], i2 = ge.indices[3 * p + 2];
    const float* P = ge.positions;
    TriRec r;
    r.v0x = P[3 * (size_t)i0]; r.v0y = P[3 * (size_t)i0 + 1]; r.v0z = P[3 * (size_t)i0 + 2];
    r.v1x = P[3 * (size_t)i1]; r.v1y = P[3 * (size_t)i1 + 1]; r.v1z = P[3 * (size_t)i1 + 2];
    r.v2x = P[3 * (size_t)i2]; r.v2y = P[3 * (size_t)i2 + 1]; r.v2z = P[3 * (size_t)i2 + 2];
    r.geo = ge.geo; r.prim = p; r.pad = ge.inst;
    tris[item_begin + i] = r;
    {   // the traversal's copy: every vertex as x y z x y (msne_device.h TriRot)
        TriRot q;
        q.v[0] = r.v0x; q.v[1] = r.v0y; q.v[2] = r.v0z; q.v[3] = r.v0x; q.v[4] = r.v0y;
        q.v[5] = r.v1x; q.v[6] = r.v1y; q.v[7] = r.v1z; q.v[8] = r.v1x; q.v[9] = r.v1y;
        q.v[10] = r.v2x; q.v[11] = r.v2y; q.v[12] = r.v2z; q.v[13] = r.v2x; q.v[14] = r.v2y;
        q.inst = ge.inst;
        rots[item_begin + i] = q;
    }
    if (attrs && (ge.normals || ge.texcoords)) {   // world.hlsl:127-149: attribute indices and the texcoords a mesh without any gets
        const uint32_t a0 = ge.indexed ? i0 : 3 * p, a1 = ge.indexed ? i1 : 3 * p + 1, a2 = ge.indexed ? i2 : 3 * p + 2;
        TriAttr t;
        t.n0x = t.n0y = t.n0z = t.n1x = t.n1y = t.n1z = t.n2x = t.n2y = t.n2z = 0.0f;
        t.t0x = 0.0f; t.t0y = 0.0f; t.t1x = 1.0f; t.t1y = 0.0f; t.t2x = 1.0f; t.t2y = 1.0f; t.pad = 0.0f;
        if (a0 >= ge.attr_count || a1 >= ge.attr_count || a2 >= ge.attr_count) { attrs[item_begin + i] = t; return; }
        if (ge.normals) {
            const float* N = ge.normals;
            t.n0x = N[3 * (size_t)a0]; t.n0y = N[3 * (size_t)a0 + 1]; t.n0z = N[3 * (size_t)a0 + 2];
            t.n1x = N[3 * (size_t)a1]; t.n1y = N[3 * (size_t)a1 + 1]; t.n1z = N[3 * (size_t)a1 + 2];
            t.n2x = N[3 * (size_t)a2]; t.n2y = N[3 * (size_t)a2 + 1]; t.n2z = N[3 * (size_t)a2 + 2];
        }
        if (ge.texcoords) {
            const float* T = ge.texcoords;
            t.t0x = T[2 * (size_t)a0]; t.t0y = T[2 * (size_t)a0 + 1]; t.t1x = T[2 * (size_t)a1]; t.t1y = T[2 * (size_t)a1 + 1]; t.t2x = T[2 * (size_t)a2]; t.t2y = T[2 * (size_t)a2 + 1];
        }
        attrs[item_begin + i] = t;
    }
}
__global__ void k_emit_items(const uint32_t* item_src, uint32_t item_begin, uint32_t n, const uint32_t* ids, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[item_begin + i] = ids[item_src[item_begin + i]];
}

// ---------------- host orchestration ----------------
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "moonshine_amd: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return false; } } while (0)

// ---------------- the top of the tree: top-down, on the host ----------------
// PLOC sees neighbours in Morton order only, and near the top of the tree that order says little about space: the last few thousand clusters are
// handed to the host, which splits them top-down by the surface-area heuristic with a full sweep over the three axes (cost of a split = area(L) x primitives(L)
// + area(R) x primitives(R)) and writes the resulting binary nodes, with their collapse tables, behind the ones PLOC made.

__global__ void k_top_gather(const uint32_t* cref, const Box* cbox, uint32_t c, BinTree t, TopCluster* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    TopCluster o;
    o.ref = cref[i]; o.box = cbox[i];
    if (o.ref & REF_LEAF) { for (int k = 0; k < 7; k++) o.cost[k] = 0.0f; o.count = 1u; }
    else { for (int k = 0; k < 7; k++) o.cost[k] = t.cost[7 * (size_t)o.ref + k]; o.count = t.count[o.ref]; }
    out[i] = o;
}

}  // namespace msne
#include "bvh_sweep.h"
namespace msne {

struct BuildScratch {
    uint32_t cap = 0;
    Box *boxes = nullptr, *sorted = nullptr, *ibox = nullptr;
    uint32_t *keys = nullptr, *keys2 = nullptr, *idx = nullptr, *idx2 = nullptr, *ghist = nullptr, *bounds = nullptr;
    uint32_t *left = nullptr, *right = nullptr, *count = nullptr; float* cost = nullptr; uint8_t* split = nullptr;
    CollapseWork *wa = nullptr, *wb = nullptr;
    uint32_t* next_count = nullptr;
    Box *cba = nullptr, *cbb = nullptr;                                       // PLOC cluster boxes (ping-pong)
    uint32_t *cra = nullptr, *crb = nullptr, *nn = nullptr, *pflags = nullptr, *bsum = nullptr; PlocState* totals = nullptr;
    uint2* bbase = nullptr;
    uint32_t *seg = nullptr, *csa = nullptr, *csb = nullptr;                   // batched builds: segment of every primitive / cluster (ping-pong)
    void* arena = nullptr; size_t arena_bytes = 0;                             // the top-down stages' working set (bvh_sweep.h), kept between builds
    bool fast_builds = false;                                                  // MsneSetBuildQuality(ctx, 0): PLOC to the roots, no sweep (an editing session's rebuilds)
    void* refit = nullptr; size_t refit_bytes = 0;                             // bvh_refit_tlas's buffers (TLAS-sized, grown on demand, kept for the context's lifetime: an
                                                                               // in-place update then allocates and frees nothing — hipFree synchronises the whole device)
    void release_arena() { if (arena) (void)hipFree(arena); arena = nullptr; arena_bytes = 0; }
    void release() {
        if (arena) (void)hipFree(arena);
        arena = nullptr; arena_bytes = 0;
        void* p[] = { boxes, sorted, ibox, keys, keys2, idx, idx2, ghist, bounds, left, right, count, cost, split, wa, wb, next_count,
                      cba, cbb, cra, crb, nn, pflags, bsum, totals, bbase, seg, csa, csb };
        for (void* q : p) if (q) (void)hipFree(q);
        const bool keep = fast_builds; void* const rf = refit; const size_t rb = refit_bytes;
        *this = BuildScratch();
        fast_builds = keep; refit = rf; refit_bytes = rb;
    }
    bool reserve(uint32_t n) {
        if (n <= cap) return true;
        release();
        const size_t N = n;
        const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
        HIPCHK(hipMalloc(&boxes, N * sizeof(Box))); HIPCHK(hipMalloc(&sorted, N * sizeof(Box))); HIPCHK(hipMalloc(&ibox, N * sizeof(Box)));
        HIPCHK(hipMalloc(&keys, N * 4)); HIPCHK(hipMalloc(&keys2, N * 4)); HIPCHK(hipMalloc(&idx, N * 4)); HIPCHK(hipMalloc(&idx2, N * 4));
        HIPCHK(hipMalloc(&ghist, ((size_t)ntiles + 1) * 256 * 4)); HIPCHK(hipMalloc(&bounds, 6 * 4));
        HIPCHK(hipMalloc(&left, N * 4)); HIPCHK(hipMalloc(&right, N * 4)); HIPCHK(hipMalloc(&count, N * 4)); HIPCHK(hipMalloc(&cost, N * 28)); HIPCHK(hipMalloc(&split, N * 8));
        HIPCHK(hipMalloc(&wa, N * sizeof(CollapseWork))); HIPCHK(hipMalloc(&wb, N * sizeof(CollapseWork)));
        HIPCHK(hipMalloc(&next_count, 4));
        const size_t nb = (N + PLOC_BLOCK - 1) / PLOC_BLOCK;
        HIPCHK(hipMalloc(&cba, N * sizeof(Box))); HIPCHK(hipMalloc(&cbb, N * sizeof(Box)));
        HIPCHK(hipMalloc(&cra, N * 4)); HIPCHK(hipMalloc(&crb, N * 4)); HIPCHK(hipMalloc(&nn, N * 4)); HIPCHK(hipMalloc(&pflags, N * 4));
        HIPCHK(hipMalloc(&bsum, nb * 4)); HIPCHK(hipMalloc(&bbase, nb * sizeof(uint2))); HIPCHK(hipMalloc(&totals, 2 * sizeof(PlocState)));
        HIPCHK(hipMalloc(&seg, N * 4)); HIPCHK(hipMalloc(&csa, N * 4)); HIPCHK(hipMalloc(&csb, N * 4));
        if (debug_poison()) {   // tests: scratch starts as garbage, as recycled device memory does in a long-lived process — an entry read before it is written shows
            struct { void* p; size_t b; } all[] = { { boxes, N * sizeof(Box) }, { sorted, N * sizeof(Box) }, { ibox, N * sizeof(Box) }, { keys, N * 4 }, { keys2, N * 4 }, { idx, N * 4 }, { idx2, N * 4 },
                { left, N * 4 }, { right, N * 4 }, { count, N * 4 }, { cost, N * 28 }, { split, N * 8 }, { wa, N * sizeof(CollapseWork) }, { wb, N * sizeof(CollapseWork) }, { cba, N * sizeof(Box) }, { cbb, N * sizeof(Box) },
                { cra, N * 4 }, { crb, N * 4 }, { nn, N * 4 }, { pflags, N * 4 }, { seg, N * 4 }, { csa, N * 4 }, { csb, N * 4 } };
            for (auto& a : all) HIPCHK(hipMemset(a.p, 0xCD, a.b));
            HIPCHK(hipDeviceSynchronize());   // (the library's streams do not wait for the null stream)
        }
        cap = n;
        return true;
    }
};

// The scratch belongs to ONE context (allocated on its device, used on its stream under its mutex): contexts that rebuild
// concurrently from different threads, or live on different GPUs, share nothing.
BuildScratch* bvh_scratch_create() { return new (std::nothrow) BuildScratch(); }
void bvh_scratch_destroy(BuildScratch* s) { if (s) { s->release(); if (s->refit) (void)hipFree(s->refit); delete s; } }
void bvh_scratch_release_arena(BuildScratch* s) { if (s) s->release_arena(); }
size_t bvh_scratch_arena_bytes(const BuildScratch* s) { return s ? s->arena_bytes : 0; }
void bvh_scratch_release(BuildScratch* s) { if (s) s->release(); }
size_t bvh_scratch_capacity(const BuildScratch* s) { return s ? s->cap : 0; }
void bvh_scratch_set_fast(BuildScratch* s, bool fast) { if (s) s->fast_builds = fast; }

// How many clusters PLOC leaves for the top-down stages (0: none, PLOC merges down to the roots; >= n: no PLOC at all, ONE sweep over every primitive).  $MSNE_SAH_TOP overrides.
// The sweep over everything makes the best trees (S1 5870 against 5770 Mrays/s with 4096 clusters and 5627 with PLOC alone, S2 3423 / 3391 / 3335, a 16 M-triangle
// scene 3848 / 3808: profiles/r04_sah_top_sweep.txt) and on the GPU it is cheap enough to be the rule: 10 ms for a million triangles (7 with clusters), 148 ms for
// 16 M (94) — the reference asks for prefer_fast_trace builds everywhere (Accel.zig:112,259,445,645).  Beyond 32 M primitives in one build PLOC goes first again
// (the sweep keeps ~100 B per position).
static uint32_t top_clusters(uint32_t n, uint32_t nseg, bool fast) {
    if (fast) return 0u;
    static const long long forced = [] { const char* e = getenv("MSNE_SAH_TOP"); return e ? atoll(e) : -1ll; }();
    if (forced >= 0) { const uint32_t m = (uint32_t)std::min<long long>(forced, 0x7fffffffll); return (m < 2 || nseg > m / 2) ? 0u : m; }
    if (n <= (32u << 20)) return n;
    const uint32_t m = std::min<uint32_t>(std::max<uint32_t>(4096u, 4u * nseg), 32768u);   // ~4 clusters per tree of a batch; beyond 16384 trees PLOC builds them whole
    return nseg > m / 2 ? 0u : m;
}

// ---- the top-down stages, sequentially on the host: the restatement the GPU stages are compared with ($MSNE_TOPDOWN=host; tests only) ----
static bool sweep_on_host(BuildScratch& S, hipStream_t s, uint32_t n, uint32_t nseg, uint32_t c, uint32_t node_base, const uint32_t* ra, const Box* ba, const uint32_t* sa,
                          BinTree t, bool rebuild_wanted, uint32_t* root_refs, Box* root_boxes) {
    const bool segmented = nseg > 1;
    TopCluster* dtop = nullptr;
    HIPCHK(hipMalloc(&dtop, (size_t)c * sizeof(TopCluster)));
    struct FreeTop { TopCluster* p; ~FreeTop() { (void)hipFree(p); } } free_top{ dtop };
    hipLaunchKernelGGL(k_top_gather, dim3((c + 255) / 256), dim3(256), 0, s, ra, ba, c, t, dtop);
    std::vector<TopCluster> cl(c); std::vector<uint32_t> cseg_h(segmented ? c : 0u);
    HIPCHK(hipMemcpyAsync(cl.data(), dtop, (size_t)c * sizeof(TopCluster), hipMemcpyDeviceToHost, s));
    if (segmented) HIPCHK(hipMemcpyAsync(cseg_h.data(), sa, (size_t)c * 4, hipMemcpyDeviceToHost, s));
    const uint32_t total = node_base + (c - nseg);   // binary nodes when all is done
    if (total > n) { fprintf(stderr, "moonshine_amd: binary node count out of range\n"); return false; }
    HostTree T;
    T.alloc(total);
    const bool rebuild_bottom = rebuild_wanted && node_base;
    RawArray<Box> prim_box;
    if (rebuild_bottom) {
        prim_box.alloc(n);
        HIPCHK(hipMemcpyAsync(T.left.data(), t.left, (size_t)node_base * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(T.right.data(), t.right, (size_t)node_base * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(prim_box.data(), S.sorted, (size_t)n * sizeof(Box), hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    if (rebuild_bottom) {
        // every PLOC cluster is rebuilt top-down over its own primitives (in primitive order), in the node ids it had
        std::atomic<bool> failed{ false };
        msne_host::parallel_for(c, [&](uint32_t i) { try {
            TopCluster& k = cl[i];
            if (k.ref & REF_LEAF) return;
            std::vector<uint32_t> ids, prims, stack{ k.ref }; std::vector<TopCluster> el;
            ids.reserve(k.count); prims.reserve(k.count); el.reserve(k.count);
            while (!stack.empty()) {
                const uint32_t r = stack.back(); stack.pop_back();
                if (r & REF_LEAF) prims.push_back(r & ~REF_LEAF);
                else { ids.push_back(r); stack.push_back(T.right[r]); stack.push_back(T.left[r]); }
            }
            std::sort(ids.begin(), ids.end()); std::sort(prims.begin(), prims.end());
            for (uint32_t pr : prims) { TopCluster e; e.ref = REF_LEAF | pr; e.box = prim_box[pr]; for (int q = 0; q < 7; q++) e.cost[q] = 0.0f; e.count = 1u; el.push_back(e); }
            TopDown td(T, ids.data());
            const TopDown::Sub r = td.run(el.data(), (uint32_t)el.size());
            k.ref = r.ref; k.box = r.box; for (int q = 0; q < 7; q++) k.cost[q] = r.cost[q];
        } catch (const std::exception&) { failed = true; } });
        if (failed) { fprintf(stderr, "moonshine_amd: out of host memory in the BVH builder\n"); return false; }
    }
    std::vector<uint32_t> top_ids(c - nseg);
    for (uint32_t i = 0; i < c - nseg; i++) top_ids[i] = node_base + i;
    TopDown top(T, top_ids.data());
    for (uint32_t a = 0, j = 0; a < c; j++) {   // the clusters of a segment are contiguous
        uint32_t b = a + 1;
        while (segmented && b < c && cseg_h[b] == cseg_h[a]) b++;
        if (!segmented) b = c;
        if (j >= nseg || (segmented && cseg_h[a] != j)) { fprintf(stderr, "moonshine_amd: builder lost a segment\n"); return false; }
        const TopDown::Sub r = top.run(cl.data() + a, b - a);
        root_refs[j] = r.ref; root_boxes[j] = r.box;
        a = b;
    }
    const uint32_t first = rebuild_bottom ? 0u : node_base, made = total - first;   // the range of ids the host wrote
    if (made) {
        HIPCHK(hipMemcpyAsync(t.left + first, T.left.data() + first, (size_t)made * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(t.right + first, T.right.data() + first, (size_t)made * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(t.box + first, T.box.data() + first, (size_t)made * sizeof(Box), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(t.cost + 7 * (size_t)first, T.cost.data() + 7 * (size_t)first, (size_t)made * 28, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(t.split + 8 * (size_t)first, T.split.data() + 8 * (size_t)first, (size_t)made * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));   // (pageable host vectors about to go out of scope)
    }
    return true;
}

// ---- the top-down stages on the GPU (kernels: bvh_sweep.h) ----
// stable LSD radix sort of (key, value) pairs on the low `bits` bits, 8 per pass; the sorted pairs end in (ka, va) — the arrays are swapped as the passes go
static void sweep_radix(hipStream_t s, uint32_t*& ka, uint32_t*& kb, uint32_t*& va, uint32_t*& vb, uint32_t n, int bits, uint32_t* ghist) {
    const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(ntiles), dim3(256), 0, s, ka, n, shift, ghist, ntiles);
        hipLaunchKernelGGL(k_radix_scan_digit, dim3(256), dim3(256), 0, s, ghist, ntiles, ghist + (size_t)ntiles * 256u);
        hipLaunchKernelGGL(k_radix_scatter, dim3(ntiles), dim3(64), 0, s, ka, va, n, shift, ghist, ghist + (size_t)ntiles * 256u, ntiles, kb, vb);
        std::swap(ka, kb); std::swap(va, vb);
    }
}
static int bits_for(uint32_t values) { int b = 0; while ((1ull << b) < values) b++; return b; }

static bool sweep_on_gpu(BuildScratch& S, hipStream_t s, uint32_t n, uint32_t nseg, uint32_t c, uint32_t node_base, const uint32_t* ra, const Box* ba, const uint32_t* sa,
                         BinTree t, bool rebuild_wanted, uint32_t* root_refs, Box* root_boxes, bool* no_memory) {
    static const bool timing = getenv("MSNE_BUILD_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    double t_ms[4] = { 0, 0, 0, 0 };   // ownership + element sort, levels, bottom-up, roots
    auto lap = [&](int k) { if (!timing) return; (void)hipStreamSynchronize(s); const auto now = std::chrono::steady_clock::now(); t_ms[k] += std::chrono::duration<double, std::milli>(now - t_prev).count(); t_prev = now; };
    const bool segmented = nseg > 1;
    const bool rebuild_bottom = rebuild_wanted && node_base;
    const uint32_t P0 = rebuild_bottom ? n : 0u, ngc = rebuild_bottom ? c : 0u, N = P0 + c, G = ngc + nseg;
    if ((uint64_t)n + c >= 0x7fffff00ull) { fprintf(stderr, "moonshine_amd: too many primitives for the top-down stages\n"); return false; }
    const uint32_t ntile = (N + SW_TILE - 1) / SW_TILE;
    const uint32_t nsort = std::max(N, std::max(P0, node_base));
    // one arena, carved
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t o_ebox = carve((size_t)N * sizeof(Box)), o_ecnt = carve((size_t)N * 4), o_eref = carve((size_t)N * 4), o_egrp = carve((size_t)N * 4);
    size_t o_ord[2][3], o_segA[2], o_segB[2], o_segR[2], o_best[2], o_act[2], o_vf[3], o_vb[3], o_cf[3], o_vf2[3], o_vb2[3], o_cf2[3], o_list[2];
    const uint32_t nsuper = (ntile + SW_SUPER - 1) / SW_SUPER;
    for (int h = 0; h < 2; h++) { for (int k = 0; k < 3; k++) o_ord[h][k] = carve((size_t)N * 4); o_segA[h] = carve((size_t)N * 4); o_segB[h] = carve((size_t)N * 4); o_segR[h] = carve((size_t)N * 4); o_best[h] = carve((size_t)N * 8); o_act[h] = carve(ntile); }
    for (int k = 0; k < 3; k++) { o_vf[k] = carve((size_t)ntile * sizeof(SwAgg)); o_vb[k] = carve((size_t)ntile * sizeof(SwAgg)); o_cf[k] = carve((size_t)ntile * 4);
                                  o_vf2[k] = carve((size_t)nsuper * sizeof(SwAgg)); o_vb2[k] = carve((size_t)nsuper * sizeof(SwAgg)); o_cf2[k] = carve((size_t)nsuper * 4); }
    const size_t o_right = carve(N);
    o_list[0] = carve((size_t)std::max(P0, 1u) * 4); o_list[1] = carve((size_t)c * 4);
    const size_t o_counts = carve(2 * 4), o_lfirst = carve((size_t)2 * (SW_MAX_LEVELS + 1) * 4), o_cid = carve((size_t)std::max(node_base, 1u) * 4);
    const size_t o_ka = carve((size_t)nsort * 4), o_kb = carve((size_t)nsort * 4), o_va = carve((size_t)nsort * 4), o_vbs = carve((size_t)nsort * 4);
    const size_t o_ghist = carve(((size_t)((nsort + RS_TILE - 1) / RS_TILE) + 1) * 256 * 4);
    const size_t o_lc = carve((size_t)std::max(P0, 1u) * 4), o_pl = carve((size_t)std::max(P0, 1u) * 4), o_nc = carve((size_t)std::max(node_base, 1u) * 4), o_pn = carve((size_t)std::max(node_base, 1u) * 4), o_no = carve((size_t)std::max(node_base, 1u) * 4);
    const size_t o_psort = carve((size_t)std::max(P0, 1u) * 4), o_pgrp = carve((size_t)std::max(P0, 1u) * 4);
    const size_t o_gsize = carve(((size_t)G + 1) * 4), o_rref = carve((size_t)nseg * 4), o_rbox = carve((size_t)nseg * sizeof(Box));
    static const size_t arena_limit = [] { const char* e = getenv("MSNE_SWEEP_ARENA_LIMIT"); return e ? (size_t)atoll(e) : ~(size_t)0; }();   // (tests: pretend the working set does not fit)
    if (off > S.arena_bytes) {
        if (S.arena) (void)hipFree(S.arena);
        S.arena = nullptr; S.arena_bytes = 0;
        if (off > arena_limit || hipMalloc(&S.arena, off) != hipSuccess) {   // nothing of the tree has been touched yet: the caller lets PLOC finish the build
            (void)hipGetLastError(); S.arena = nullptr;
            if (no_memory) *no_memory = true;
            return false;
        }
        S.arena_bytes = off;
    }
    char* A = (char*)S.arena;
    static const bool poison = debug_poison();   // tests: the working set starts as garbage, as it does in a long-lived process
    if (poison) HIPCHK(hipMemsetAsync(A, 0xCD, off, s));
    auto U32 = [&](size_t o) { return (uint32_t*)(A + o); };
    SwState St{};
    St.N = N; St.P0 = P0; St.ngc = ngc; St.node_base = node_base; St.ntile = ntile;
    Box* ebox = (Box*)(A + o_ebox); uint32_t *ecnt = U32(o_ecnt), *eref = U32(o_eref), *egrp = U32(o_egrp);
    St.ebox = ebox; St.ecnt = ecnt; St.eref = eref; St.egrp = egrp; St.cid = U32(o_cid);
    for (int h = 0; h < 2; h++) { for (int k = 0; k < 3; k++) St.ord[h][k] = U32(o_ord[h][k]); St.segA[h] = U32(o_segA[h]); St.segB[h] = U32(o_segB[h]); St.segR[h] = U32(o_segR[h]); St.best[h] = (unsigned long long*)(A + o_best[h]); St.tile_act[h] = (uint8_t*)(A + o_act[h]); }
    for (int k = 0; k < 3; k++) { St.vf[k] = (SwAgg*)(A + o_vf[k]); St.vb[k] = (SwAgg*)(A + o_vb[k]); St.cf[k] = U32(o_cf[k]); St.vf2[k] = (SwAgg*)(A + o_vf2[k]); St.vb2[k] = (SwAgg*)(A + o_vb2[k]); St.cf2[k] = U32(o_cf2[k]); }
    St.right_side = (uint8_t*)(A + o_right); St.list[0] = U32(o_list[0]); St.list[1] = U32(o_list[1]); St.list_count = U32(o_counts); St.level_first = U32(o_lfirst);
    uint32_t *ka = U32(o_ka), *kb = U32(o_kb), *va = U32(o_va), *vb = U32(o_vbs), *ghist = U32(o_ghist), *gsize = U32(o_gsize);
    auto grid = [](uint32_t m) { return dim3((m + 255) / 256); };

    // 1. which cluster owns which primitive / PLOC node; primitives and node ids grouped by cluster (stable: ascending inside)
    uint32_t *prim_sorted = U32(o_psort), *prim_group = U32(o_pgrp);
    if (rebuild_bottom) {
        uint32_t *leaf_cluster = U32(o_lc), *parent_leaf = U32(o_pl), *node_cluster = U32(o_nc), *parent_node = U32(o_pn), *node_owner = U32(o_no);
        HIPCHK(hipMemsetAsync(leaf_cluster, 0xFF, (size_t)n * 4, s));
        HIPCHK(hipMemsetAsync(node_cluster, 0xFF, (size_t)node_base * 4, s));
        hipLaunchKernelGGL(k_sw_roots, grid(c), dim3(256), 0, s, ra, c, leaf_cluster, node_cluster);
        hipLaunchKernelGGL(k_sw_parents, grid(node_base), dim3(256), 0, s, t, node_base, parent_leaf, parent_node);
        hipLaunchKernelGGL(k_sw_owner, grid(n + node_base), dim3(256), 0, s, n, node_base, parent_leaf, parent_node, leaf_cluster, node_cluster, node_owner);
        const int cb = bits_for(c);
        HIPCHK(hipMemcpyAsync(ka, leaf_cluster, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_sw_iota, grid(n), dim3(256), 0, s, n, va);
        sweep_radix(s, ka, kb, va, vb, n, cb, ghist);
        HIPCHK(hipMemcpyAsync(prim_group, ka, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(prim_sorted, va, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(ka, node_owner, (size_t)node_base * 4, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_sw_iota, grid(node_base), dim3(256), 0, s, node_base, va);
        sweep_radix(s, ka, kb, va, vb, node_base, cb, ghist);
        HIPCHK(hipMemcpyAsync(U32(o_cid), va, (size_t)node_base * 4, hipMemcpyDeviceToDevice, s));
    }
    // 2. elements, groups, the three orders
    hipLaunchKernelGGL(k_sw_elements, grid(N), dim3(256), 0, s, N, P0, ngc, prim_sorted, prim_group, S.sorted, ra, ba, segmented ? sa : nullptr, t, ebox, ecnt, eref, egrp);
    if (ngc) {
        hipLaunchKernelGGL(k_sw_cluster_sizes, grid(c + 1u), dim3(256), 0, s, ra, c, t, gsize);
        hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, s, gsize, ngc + 1u);   // in place: first position of every cluster's primitives, gsize[ngc] = P0
    }
    hipLaunchKernelGGL(k_sw_tree_first, grid(c + 1u), dim3(256), 0, s, segmented ? sa : nullptr, c, ngc, nseg, P0, gsize);   // gsize[G] = N
    const int gb = G > 1 ? bits_for(G) : 0;
    for (int axis = 0; axis < 3; axis++) {
        hipLaunchKernelGGL(k_sw_keys, grid(N), dim3(256), 0, s, N, ebox, axis, ka, va);
        sweep_radix(s, ka, kb, va, vb, N, 32, ghist);
        if (gb) { hipLaunchKernelGGL(k_gather_u32, grid(N), dim3(256), 0, s, egrp, va, N, ka); sweep_radix(s, ka, kb, va, vb, N, gb, ghist); }   // stable: back into groups, sorted inside
        HIPCHK(hipMemcpyAsync(St.ord[0][axis], va, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    }
    HIPCHK(hipMemsetAsync(St.list_count, 0, 8, s));
    hipLaunchKernelGGL(k_sw_init, dim3(ntile), dim3(SW_TILE), 0, s, St, gsize);
    lap(0);
    // 3. one tree level per pass, every build at once
    const uint32_t want[2] = { rebuild_bottom ? n - c : 0u, c - nseg };
    std::vector<uint32_t> lfirst((size_t)2 * (SW_MAX_LEVELS + 1));
    uint32_t made[2] = { 0, 0 }, level = 0;
    int cur = 0;
    for (;;) {
        for (uint32_t g = 0; g < 4u && level < SW_MAX_LEVELS; g++, level++) {
            hipLaunchKernelGGL(k_sw_agg, dim3(ntile, 3), dim3(SW_TILE), 0, s, St, cur, level);
            if (level < MAX_SWEEP_DEPTH) {
                if (nsuper > 1) hipLaunchKernelGGL(k_sw_agg2, dim3(nsuper, 3), dim3(64), 0, s, St, cur);
                hipLaunchKernelGGL(k_sw_cost, dim3(ntile, 3), dim3(SW_TILE), 0, s, St, cur);
            }
            hipLaunchKernelGGL(k_sw_split, dim3(ntile), dim3(SW_TILE), 0, s, St, cur, t);
            hipLaunchKernelGGL(k_sw_pagg, dim3(ntile, 3), dim3(SW_TILE), 0, s, St, cur);
            if (nsuper > 1) hipLaunchKernelGGL(k_sw_pagg2, dim3(nsuper, 3), dim3(64), 0, s, St, cur);
            hipLaunchKernelGGL(k_sw_part, dim3(ntile, 3), dim3(SW_TILE), 0, s, St, cur, t);
            cur ^= 1;
        }
        HIPCHK(hipMemcpyAsync(made, St.list_count, 8, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (made[0] == want[0] && made[1] == want[1]) break;
        if (made[0] > want[0] || made[1] > want[1] || level >= SW_MAX_LEVELS) { fprintf(stderr, "moonshine_amd: the top-down stages lost count of their nodes (%u / %u, %u / %u after %u levels)\n", made[0], want[0], made[1], want[1], level); return false; }
    }
    HIPCHK(hipMemcpyAsync(lfirst.data(), St.level_first, lfirst.size() * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    lap(1);
    // 4. boxes and collapse tables, deepest level first; the cluster trees, then the top trees over them
    for (uint32_t kind = 0; kind < 2; kind++)
        for (uint32_t L = level; L-- > 0;) {
            const uint32_t from = lfirst[(size_t)kind * (SW_MAX_LEVELS + 1) + L], to = L + 1 < level ? lfirst[(size_t)kind * (SW_MAX_LEVELS + 1) + L + 1] : made[kind];
            if (to > from) hipLaunchKernelGGL(k_sw_up, grid(to - from), dim3(256), 0, s, St.list[kind] + from, to - from, t, S.sorted);
        }
    lap(2);
    hipLaunchKernelGGL(k_sw_tree_roots, grid(nseg), dim3(256), 0, s, St, gsize, nseg, t, U32(o_rref), (Box*)(A + o_rbox));
    HIPCHK(hipMemcpyAsync(root_refs, A + o_rref, (size_t)nseg * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(root_boxes, A + o_rbox, (size_t)nseg * sizeof(Box), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    lap(3);
    if (timing && n >= 4096u) fprintf(stderr, "moonshine_amd builder: %u primitives in %u trees, %u clusters swept top-down on the GPU over %u levels: ownership + sorts %.2f ms, levels %.2f ms, tables %.2f ms, roots %.2f ms\n",
                                      n, nseg, c, level, t_ms[0], t_ms[1], t_ms[2], t_ms[3]);
    return true;
}

// Builds `nseg` wide BVHs over the n boxes in S.boxes — segment j = boxes [seg_first[j], seg_first[j + 1]), none empty; nseg == 1: one tree over
// all of them.  Nodes are appended at *node_counter (device), items at *item_counter; item_src[pos] = source box index for final item position pos.
// Returns the root node index and the root box of every segment (host).
static bool build_from_boxes(BuildScratch& S, hipStream_t s, uint32_t n, uint32_t nseg, const uint32_t* seg_first /* host, nseg + 1 */, Node8* nodes, uint32_t* node_counter, uint32_t node_capacity,
                             uint32_t* item_counter, uint32_t* item_src, uint32_t* roots_out, Box* root_boxes) try {
    static const bool phase_timing = getenv("MSNE_BUILD_TIMING") != nullptr;
    auto ph_prev = std::chrono::steady_clock::now(); double ph_ms[4] = { 0, 0, 0, 0 };   // sort, PLOC, host stages, collapse
    auto phase = [&](int k) { if (!phase_timing) return; (void)hipStreamSynchronize(s); const auto now = std::chrono::steady_clock::now(); ph_ms[k] += std::chrono::duration<double, std::milli>(now - ph_prev).count(); ph_prev = now; };
    const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
    const bool segmented = nseg > 1;
    uint32_t* d_segfirst = nullptr; uint32_t* d_bounds = S.bounds;
    struct Guard { uint32_t*& a; uint32_t*& b; uint32_t* keep; ~Guard() { if (a) (void)hipFree(a); if (b && b != keep) (void)hipFree(b); } } guard{ d_segfirst, d_bounds, S.bounds };
    if (segmented) {
        HIPCHK(hipMalloc(&d_segfirst, ((size_t)nseg + 1) * 4));
        d_bounds = nullptr;
        HIPCHK(hipMalloc(&d_bounds, (size_t)nseg * 24));
        HIPCHK(hipMemcpyAsync(d_segfirst, seg_first, ((size_t)nseg + 1) * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_seg_of, dim3((n + 255) / 256), dim3(256), 0, s, d_segfirst, nseg, n, S.seg);
        hipLaunchKernelGGL(k_fill_bounds, dim3((6 * nseg + 255) / 256), dim3(256), 0, s, d_bounds, nseg);
        hipLaunchKernelGGL(k_bounds_seg, dim3((n + 255) / 256), dim3(256), 0, s, S.boxes, S.seg, n, d_bounds);
    } else {
        const uint32_t init_bounds[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u };
        HIPCHK(hipMemcpyAsync(S.bounds, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_bounds, dim3(std::min<uint32_t>((n + 255) / 256, 1024)), dim3(256), 0, s, S.boxes, n, S.bounds);
    }
    // 21 bits per axis where 10 run out of cells; up to 4 M primitives the coarse code is kept — PLOC then sees primitives of one cell in input order, i.e. in the mesh's own
    // order, which on S1 gives slightly better trees (13.48 against 13.82 node visits per closest-hit ray, +0.7 % Mrays/s) and saves four sort passes
    static const int morton_bits = [] { const char* e = getenv("MSNE_MORTON_BITS"); return e ? atoi(e) : 0; }();
    const bool fine = morton_bits ? morton_bits > 10 : n > (1u << 22);
    hipLaunchKernelGGL(k_morton, dim3((n + 255) / 256), dim3(256), 0, s, S.boxes, n, d_bounds, segmented ? S.seg : nullptr, fine ? 2097152.0f : 1024.0f, S.keys, S.nn /* high words, by primitive (free until PLOC) */, S.idx);
    uint32_t *ka = S.keys, *kb = S.keys2, *va = S.idx, *vb = S.idx2;
    auto radix_pass = [&](int shift) {
        hipLaunchKernelGGL(k_radix_hist, dim3(ntiles), dim3(256), 0, s, ka, n, shift, S.ghist, ntiles);
        hipLaunchKernelGGL(k_radix_scan_digit, dim3(256), dim3(256), 0, s, S.ghist, ntiles, S.ghist + (size_t)ntiles * 256u);
        hipLaunchKernelGGL(k_radix_scatter, dim3(ntiles), dim3(64), 0, s, ka, va, n, shift, S.ghist, S.ghist + (size_t)ntiles * 256u, ntiles, kb, vb);
        std::swap(ka, kb); std::swap(va, vb);
    };
    for (int pass = 0; pass < 4; pass++) radix_pass(pass * 8);                       // low word
    if (fine) {
        hipLaunchKernelGGL(k_gather_u32, dim3((n + 255) / 256), dim3(256), 0, s, S.nn, va, n, ka);   // the sort is stable: four more passes on the high word, in the order reached so far
        for (int pass = 0; pass < 4; pass++) radix_pass(pass * 8);
    }
    if (segmented) {   // the sort is stable: further passes on the segment number bring every segment back together, in Morton order inside
        int bits = 0; while ((1ull << bits) < nseg) bits++;
        hipLaunchKernelGGL(k_gather_u32, dim3((n + 255) / 256), dim3(256), 0, s, S.seg, va, n, ka);
        for (int shift = 0; shift < bits; shift += 8) radix_pass(shift);
    }
    hipLaunchKernelGGL(k_gather_boxes, dim3((n + 255) / 256), dim3(256), 0, s, S.boxes, va, n, S.sorted);
    phase(0);
    BinTree t{ S.left, S.right, S.ibox, S.cost, S.split, S.count };
    std::vector<uint32_t> root_refs(nseg);
    if (n >= 2) {
        HIPCHK(hipMemcpyAsync(S.cba, S.sorted, (size_t)n * sizeof(Box), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_ploc_init, dim3((n + 255) / 256), dim3(256), 0, s, n, S.cra);
        if (segmented) hipLaunchKernelGGL(k_gather_u32, dim3((n + 255) / 256), dim3(256), 0, s, S.seg, va, n, S.csa);
        static const uint32_t radius = [] { const char* e = getenv("MSNE_PLOC_RADIUS"); return e ? (uint32_t)atoi(e) : (uint32_t)PLOC_RADIUS; }();
        uint32_t *ra = S.cra, *rb = S.crb; Box *ba = S.cba, *bb = S.cbb;
        uint32_t *sa = segmented ? S.csa : nullptr, *sb = segmented ? S.csb : nullptr;
        // Rounds are queued in groups of PLOC_GROUP without a host round trip: every kernel reads the cluster count of its round from
        // the device (slot r & 1 of S.totals) and runs on a grid sized for the count at the start of the group; a round that starts
        // with one cluster per segment copies them through unchanged.  One read-back per group instead of one per round (1 M triangles: ~56 rounds).
        constexpr uint32_t PLOC_GROUP = 8;
        PlocState st0{ n, 0u, 0u, 0u };
        HIPCHK(hipMemcpyAsync(S.totals, &st0, sizeof st0, hipMemcpyHostToDevice, s));
        uint32_t c = n, round = 0, node_base = 0;
        // (each round at most halves the clusters: `group` rounds from c leave at least c >> group)
        uint32_t stop = std::max(nseg, top_clusters(n, nseg, S.fast_builds));
        auto ploc_down_to = [&](uint32_t stop) -> bool {
        while (c > stop) {
            const uint32_t nb = (c + PLOC_BLOCK - 1) / PLOC_BLOCK;
            uint32_t group = PLOC_GROUP;
            if (stop > nseg) { group = 1; while (group < PLOC_GROUP && (c >> (group + 1)) >= stop) group++; }
            for (uint32_t g = 0; g < group; g++, round++) {
                const PlocState* cur = S.totals + (round & 1u); PlocState* nxt = S.totals + ((round + 1u) & 1u);
                hipLaunchKernelGGL(k_ploc_nn, dim3(nb), dim3(PLOC_BLOCK), 0, s, ba, sa, cur, radius, S.nn);
                hipLaunchKernelGGL(k_ploc_mark, dim3(nb), dim3(PLOC_BLOCK), 0, s, S.nn, cur, S.pflags, S.bsum);
                hipLaunchKernelGGL(k_ploc_scan, dim3(1), dim3(1024), 0, s, S.bsum, cur, nxt, S.bbase, nseg);
                hipLaunchKernelGGL(k_ploc_merge, dim3(nb), dim3(PLOC_BLOCK), 0, s, ra, ba, sa, S.nn, S.pflags, S.bbase, cur, t, rb, bb, sb);
                std::swap(ra, rb); std::swap(ba, bb); std::swap(sa, sb);
            }
            PlocState now{};
            HIPCHK(hipMemcpyAsync(&now, S.totals + (round & 1u), sizeof now, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (now.stuck || now.c >= c || now.c < nseg) { fprintf(stderr, "moonshine_amd: PLOC made no progress\n"); return false; }
            c = now.c; node_base = now.node_base;
        }
        return true; };
        if (!ploc_down_to(stop)) return false;
        phase(1);
        bool swept = false;
        if (c > nseg) {   // the rest top-down: a surface-area sweep over every cluster's primitives and over the clusters themselves (bvh_sweep.h)
            static const bool rebuild_wanted = [] { const char* e = getenv("MSNE_SAH_BOTTOM"); return e ? atoi(e) != 0 : true; }();
            const char* where = getenv("MSNE_TOPDOWN");                        // "host": the sequential restatement (tests); read at every build
            const bool on_host = where && strcmp(where, "host") == 0;
            if (on_host) { if (!sweep_on_host(S, s, n, nseg, c, node_base, ra, ba, sa, t, rebuild_wanted, root_refs.data(), root_boxes)) return false; }
            else {
                bool no_memory = false;
                if (!sweep_on_gpu(S, s, n, nseg, c, node_base, ra, ba, sa, t, rebuild_wanted, root_refs.data(), root_boxes, &no_memory)) {
                    if (!no_memory) return false;
                    // the sweep's working set (~100 B per position) does not fit: PLOC carries on to the roots — a slightly worse tree instead of a failed build
                    fprintf(stderr, "moonshine_amd: no device memory for the top-down stages of a %u-primitive build; PLOC builds it whole\n", n);
                    stop = nseg;
                    if (!ploc_down_to(stop)) return false;
                } else swept = true;
            }
            if (on_host) swept = true;
        }
        if (!swept) {   // PLOC went down to one cluster per tree
            HIPCHK(hipMemcpyAsync(root_refs.data(), ra, (size_t)nseg * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(root_boxes, ba, (size_t)nseg * sizeof(Box), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
        }
    } else {
        root_refs[0] = REF_LEAF | 0u;
        HIPCHK(hipMemcpyAsync(root_boxes, S.sorted, sizeof(Box), hipMemcpyDeviceToHost, s));
    }
    phase(2);
    // one wide root node per segment
    uint32_t root_wide = 0;
    HIPCHK(hipMemcpyAsync(&root_wide, node_counter, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if ((uint64_t)root_wide + n + nseg > node_capacity) { fprintf(stderr, "moonshine_amd: node pool exhausted\n"); return false; }
    const uint32_t after = root_wide + nseg;
    HIPCHK(hipMemcpyAsync(node_counter, &after, 4, hipMemcpyHostToDevice, s));
    std::vector<CollapseWork> w0(nseg);
    for (uint32_t j = 0; j < nseg; j++) { w0[j] = CollapseWork{ root_refs[j], root_wide + j }; roots_out[j] = root_wide + j; }
    HIPCHK(hipMemcpyAsync(S.wa, w0.data(), (size_t)nseg * sizeof(CollapseWork), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));   // (w0 is pageable host memory about to go out of scope)
    uint32_t nwork = nseg;
    CollapseWork *cur = S.wa, *nxt = S.wb;
    while (nwork) {
        HIPCHK(hipMemsetAsync(S.next_count, 0, 4, s));
        hipLaunchKernelGGL(k_collapse, dim3((nwork + 63) / 64), dim3(64), 0, s, cur, nwork, nxt, S.next_count, t, S.sorted, va,
                           nodes, node_counter, item_counter, item_src);
        HIPCHK(hipMemcpyAsync(&nwork, S.next_count, 4, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::swap(cur, nxt);
    }
    phase(3);
    if (phase_timing && n >= 4096u) fprintf(stderr, "moonshine_amd builder: %u primitives: sort %.2f ms, PLOC %.2f ms, top-down stages %.2f ms, collapse %.2f ms\n", n, ph_ms[0], ph_ms[1], ph_ms[2], ph_ms[3]);
    return true;
} catch (const std::exception& e) {   // (host vectors of the top-down stages)
    fprintf(stderr, "moonshine_amd: BVH build failed on the host: %s\n", e.what());
    return false;
}

// BLASes over the triangles of geometry lists (Accel.zig:94-184; one BLAS per unique mesh list, :315-343) — ALL the BLASes a rebuild needs in one pass, as the
// reference hands them to one vkCmdBuildAccelerationStructuresKHR: `geos` lists the geometries of every job one after the other with tri_offset running
// over the whole batch, job j owns triangles [job_first[j], job_first[j + 1]).  A scene of thousands of small meshes costs one build, not thousands
// (2 000 meshes of 320 triangles: 1.65 s one by one).
bool bvh_build_blas_batch(BuildScratch* scratch, hipStream_t s, const std::vector<BlasGeo>& geos, const std::vector<uint32_t>& job_first, Node8* nodes, uint32_t* node_counter, uint32_t node_capacity,
                          TriRec* tris, TriRot* rots, TriAttr* attrs, uint32_t* tri_counter, uint32_t* item_src, uint32_t* roots_out, float* root_boxes /* 6 per job */) {
    const uint32_t njobs = (uint32_t)job_first.size() - 1u, ntris = job_first.back();
    if (njobs == 0 || ntris == 0) return true;
    if (!scratch) return false;
    BuildScratch& g_scratch = *scratch;
    if (!g_scratch.reserve(ntris)) return false;
    BlasGeo* dgeos = nullptr;
    HIPCHK(hipMalloc(&dgeos, geos.size() * sizeof(BlasGeo)));
    struct Free { BlasGeo* p; ~Free() { (void)hipFree(p); } } free_geos{ dgeos };
    HIPCHK(hipMemcpyAsync(dgeos, geos.data(), geos.size() * sizeof(BlasGeo), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_prim_boxes_tris, dim3((ntris + 255) / 256), dim3(256), 0, s, dgeos, (uint32_t)geos.size(), ntris, g_scratch.boxes);
    uint32_t item_begin = 0;
    HIPCHK(hipMemcpyAsync(&item_begin, tri_counter, 4, hipMemcpyDeviceToHost, s));
    std::vector<Box> rb(njobs);
    if (!build_from_boxes(g_scratch, s, ntris, njobs, job_first.data(), nodes, node_counter, node_capacity, tri_counter, item_src, roots_out, rb.data())) return false;
    hipLaunchKernelGGL(k_emit_tris, dim3((ntris + 255) / 256), dim3(256), 0, s, dgeos, (uint32_t)geos.size(), item_src, item_begin, ntris, tris, rots, attrs);
    HIPCHK(hipStreamSynchronize(s));
    for (uint32_t j = 0; j < njobs; j++) for (int k = 0; k < 3; k++) { root_boxes[6 * j + k] = rb[j].lo[k]; root_boxes[6 * j + 3 + k] = rb[j].hi[k]; }
    return true;
}

// ---------------- TLAS (Accel.zig:484): instance world boxes on the GPU, then the same builder ----------------
// A TLAS leaf is the world box of the instance's TRANSFORMED VERTICES, not of the transformed corners of its BLAS root box (up to 1.7x
// wider per axis under rotation; every false TLAS hit costs a change of space in the traversal).  One workgroup per instance reduces
// min / max over the vertices of its meshes; an instance without a finite vertex, or flagged `exact = 0`, takes the corner box.
struct TlasInst { float T[12]; float blas_box[6]; uint32_t mesh_begin, mesh_end, exact; float cull_pad; /* context.hip instance_cull_pad */ };   // mesh_begin..mesh_end index TlasMesh
struct TlasMesh { const float* positions; uint32_t count, pad; };

// One workgroup per instance: the box of its transformed vertices (the TLAS leaf box) and — sphere_ids != nullptr — its world-space bounding sphere into
// spheres[sphere_ids[i]]: centre = the box's, radius = the farthest transformed vertex (msne_device.h TlasLeaf; the traversal inflates it).
__global__ __launch_bounds__(256) void k_instance_boxes(const TlasInst* insts, const TlasMesh* meshes, uint32_t n, Box* boxes, const uint32_t* sphere_ids, float4* spheres) {
    __shared__ float s_lo[3][256 / 64], s_hi[3][256 / 64], s_r2[256 / 64];
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    const TlasInst in = insts[i];
    m34 T;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) T.m[r][c] = in.T[4 * r + c];
    float lo[3] = { 3e38f, 3e38f, 3e38f }, hi[3] = { -3e38f, -3e38f, -3e38f };
    if (in.exact) {
        for (uint32_t m = in.mesh_begin; m < in.mesh_end; m++) {
            const TlasMesh me = meshes[m];
            for (uint32_t v = threadIdx.x; v < me.count; v += 256) {
                const f3 q = m34_mul_point(T, F3(me.positions[3 * (size_t)v], me.positions[3 * (size_t)v + 1], me.positions[3 * (size_t)v + 2]));
                if (!(q.x == q.x && q.y == q.y && q.z == q.z)) continue;   // NaN vertices belong to inactive triangles
                lo[0] = fminf(lo[0], q.x); lo[1] = fminf(lo[1], q.y); lo[2] = fminf(lo[2], q.z);
                hi[0] = fmaxf(hi[0], q.x); hi[1] = fmaxf(hi[1], q.y); hi[2] = fmaxf(hi[2], q.z);
            }
        }
    }
    for (int k = 0; k < 3; k++) {
        for (int o = 32; o >= 1; o >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o)); }
        if ((threadIdx.x & 63) == 0) { s_lo[k][threadIdx.x >> 6] = lo[k]; s_hi[k][threadIdx.x >> 6] = hi[k]; }
    }
    __syncthreads();
    for (int k = 0; k < 3; k++) { lo[k] = s_lo[k][0]; hi[k] = s_hi[k][0]; for (int w = 1; w < 256 / 64; w++) { lo[k] = fminf(lo[k], s_lo[k][w]); hi[k] = fmaxf(hi[k], s_hi[k][w]); } }
    const bool seen = !(lo[0] > hi[0]);
    if (!seen) for (int k = 0; k < 8; k++) {   // no vertex seen: the transformed corners of the BLAS root box
        const f3 q = m34_mul_point(T, F3((k & 1) ? in.blas_box[3] : in.blas_box[0], (k & 2) ? in.blas_box[4] : in.blas_box[1], (k & 4) ? in.blas_box[5] : in.blas_box[2]));
        lo[0] = fminf(lo[0], q.x); lo[1] = fminf(lo[1], q.y); lo[2] = fminf(lo[2], q.z);
        hi[0] = fmaxf(hi[0], q.x); hi[1] = fmaxf(hi[1], q.y); hi[2] = fmaxf(hi[2], q.z);
    }
    if (sphere_ids) {   // the farthest vertex from the box's centre (second pass over the same vertices)
        const f3 c = F3(0.5f * lo[0] + 0.5f * hi[0], 0.5f * lo[1] + 0.5f * hi[1], 0.5f * lo[2] + 0.5f * hi[2]);
        float r2 = 0.0f;
        if (seen) {
            for (uint32_t m = in.mesh_begin; m < in.mesh_end; m++) {
                const TlasMesh me = meshes[m];
                for (uint32_t v = threadIdx.x; v < me.count; v += 256) {
                    const f3 q = m34_mul_point(T, F3(me.positions[3 * (size_t)v], me.positions[3 * (size_t)v + 1], me.positions[3 * (size_t)v + 2]));
                    if (!(q.x == q.x && q.y == q.y && q.z == q.z)) continue;
                    const f3 e = sub(q, c);
                    r2 = fmaxf(r2, dot(e, e));
                }
            }
        } else { const f3 e = sub(F3(hi[0], hi[1], hi[2]), c); r2 = dot(e, e); }   // the box's own circumscribed sphere
        for (int o = 32; o >= 1; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o));
        if ((threadIdx.x & 63) == 0) s_r2[threadIdx.x >> 6] = r2;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 0; w < 256 / 64; w++) r2 = fmaxf(r2, s_r2[w]);
            if (!(r2 >= 0.0f) || !(r2 < 3e38f) || !(c.x == c.x && c.y == c.y && c.z == c.z)) r2 = 3e38f;   // (not finite: a sphere nothing is outside of)
            // (the slack is a bound per coordinate: sqrt 3 of it along a diagonal.  A radius whose SQUARE the traversal could not hold — vertices beyond ~1e18 from the
            // centre, a squared distance that overflowed here — is stored as 3e38: its square is infinite there and nothing is outside of it)
            const float rad = r2 < 1e36f ? sqrtf(r2) * 1.000001f + 1.7320509f * in.cull_pad : 3.0e38f;
            spheres[sphere_ids[i]] = make_float4(c.x, c.y, c.z, rad < 1e18f ? rad : 3.0e38f);
        }
    }
    if (threadIdx.x != 0) return;
    // grown by what the world-space ray and the instance-space hit can differ by (context.hip instance_cull_pad), and by an ulp for the rounding of this very sum
    Box b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = lo[k] - in.cull_pad; b.hi[k] = hi[k] + in.cull_pad;
        b.lo[k] -= 1.2e-7f * fabsf(b.lo[k]) + 1e-30f; b.hi[k] += 1.2e-7f * fabsf(b.hi[k]) + 1e-30f;
        if (!(b.lo[k] > -3e38f)) b.lo[k] = -3e38f;
        if (!(b.hi[k] < 3e38f)) b.hi[k] = 3e38f;
    }
    boxes[i] = b;
}

// insts / meshes: host arrays describing the n visible instances (ids: instance index per entry)
bool bvh_build_tlas(BuildScratch* scratch, hipStream_t s, const TlasInst* insts, const uint32_t* host_ids, uint32_t n, const TlasMesh* meshes, uint32_t nmeshes,
                    Node8* nodes, uint32_t* node_counter, uint32_t node_capacity, uint32_t* tlas_items, uint32_t* item_counter, uint32_t* item_src, uint32_t* root_out,
                    float4* spheres /* per INSTANCE (indexed by host_ids[i]): its world-space bounding sphere */) {
    if (n == 0) { *root_out = MAX_UINT; return true; }
    if (!scratch) return false;
    BuildScratch& g_scratch = *scratch;
    if (!g_scratch.reserve(n)) return false;
    TlasInst* dinst = nullptr; TlasMesh* dmesh = nullptr; uint32_t* dids = nullptr;
    HIPCHK(hipMalloc(&dinst, (size_t)n * sizeof(TlasInst)));
    HIPCHK(hipMalloc(&dmesh, (size_t)std::max(nmeshes, 1u) * sizeof(TlasMesh)));
    HIPCHK(hipMalloc(&dids, (size_t)n * 4));
    HIPCHK(hipMemcpyAsync(dinst, insts, (size_t)n * sizeof(TlasInst), hipMemcpyHostToDevice, s));
    if (nmeshes) HIPCHK(hipMemcpyAsync(dmesh, meshes, (size_t)nmeshes * sizeof(TlasMesh), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dids, host_ids, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_instance_boxes, dim3(n), dim3(256), 0, s, dinst, dmesh, n, g_scratch.boxes, dids, spheres);
    uint32_t item_begin = 0;
    HIPCHK(hipMemcpyAsync(&item_begin, item_counter, 4, hipMemcpyDeviceToHost, s));
    Box rb;
    bool ok = build_from_boxes(g_scratch, s, n, 1u, nullptr, nodes, node_counter, node_capacity, item_counter, item_src, root_out, &rb);
    if (ok) {
        hipLaunchKernelGGL(k_emit_items, dim3((n + 255) / 256), dim3(256), 0, s, item_src, item_begin, n, dids, tlas_items);
        HIPCHK(hipStreamSynchronize(s));
    }
    (void)hipFree(dinst); (void)hipFree(dmesh); (void)hipFree(dids);
    return ok;
}

// ---------------- TLAS in-place update (Accel.zig:567-601 recordUpdateSingleTransform: vkCmdBuildAccelerationStructuresKHR in UPDATE mode) ----------------
// An instance's transform changed: its TLAS leaf gets the box of its newly transformed vertices and the boxes on the way to the root are
// re-fitted, the tree itself stays as it was built (like the reference's update, its quality degrades with large moves until the next
// rebuild).  Two link tables, derived from the finished TLAS by one pass over its nodes: parent (node, slot) of every TLAS node and of every
// TLAS leaf item.
__global__ void k_tlas_links(const Node8* nodes, uint32_t node_begin, uint32_t node_end, uint32_t item_begin, uint2* node_parent, uint2* item_parent, uint32_t root) {
    const uint32_t n = node_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= node_end) return;
    if (n == root) node_parent[n - node_begin] = make_uint2(MAX_UINT, 0u);
    const Node8 nd = nodes[n];
    uint32_t ci = 0, li = 0;
    for (uint32_t s = 0; s < 8; s++) {
        if ((nd.imask >> s) & 1u) { node_parent[nd.child_base + ci - node_begin] = make_uint2(n, s); ci++; }
        else if ((nd.lmask >> s) & 1u) { item_parent[nd.item_base + li - item_begin] = make_uint2(n, s); li++; }
    }
}

// the quantisation of k_collapse, for one child box on a node's grid; false when the box does not fit the grid (the node has to be re-gridded)
__device__ __forceinline__ bool quantise_child(const float origin[3], const uint8_t e[3], const Box& b, uint8_t ql[3], uint8_t qh[3]) {
    const float grow = box_growth(b);
    for (int k = 0; k < 3; k++) {
        const float inv_s = u2f((uint32_t)(254 - e[k]) << 23);
        const float lo = floorf((b.lo[k] - grow - origin[k]) * inv_s - 1e-3f), hi = ceilf((b.hi[k] + grow - origin[k]) * inv_s + 1e-3f);
        if (!(lo >= 0.0f) || !(hi <= 255.0f) || !(hi >= 0.0f) || !(lo <= 255.0f)) return false;   // (NaN or an empty box: does not fit any grid)
        ql[k] = (uint8_t)lo; qh[k] = (uint8_t)hi;
    }
    return true;
}
__device__ __forceinline__ Box dequantise_child(const Node8& nd, uint32_t s) {
    Box b;
    const float o[3] = { nd.ox, nd.oy, nd.oz }; const uint8_t e[3] = { nd.ex, nd.ey, nd.ez };
    for (int k = 0; k < 3; k++) {
        const float sc = u2f((uint32_t)e[k] << 23);
        b.lo[k] = o[k] + (float)nd.qlo[k][s] * sc; b.hi[k] = o[k] + (float)nd.qhi[k][s] * sc;
    }
    return b;
}

// ---- the refit, one thread per edit ----
// Pass 1 (k_tlas_refit_mark): every edit stores its leaf's new box and walks towards the root, marking nodes dirty; a node that was already dirty ends the walk (its
// ancestors are marked), a node that becomes dirty counts as one pending child of its parent.  Pass 2 (k_tlas_refit_apply): a dirty node is re-fitted ONCE, when none of
// its dirty children is pending any more — by whoever gets there first (the edits under it, or the thread that finished its last dirty child) — from its children as
// they are now: edited leaves with their new boxes, re-fitted child nodes with their new bounds, everything else as quantised.  The planes of a changed child are
// re-quantised on the node's grid; only when one does not fit is the grid re-made over the union of all children.
struct RefitState {
    uint32_t* dirty; uint32_t* pending; uint32_t* claimed;   // per TLAS node (index - node_begin), zeroed before every refit
    Box* node_bounds;                                          // per TLAS node: what a re-fitted node's parent has to bound
    uint32_t* item_edited; Box* item_box;                      // per TLAS leaf item (index - item_begin): the item was edited / its new box
};

__global__ void k_tlas_refit_mark(RefitState R, uint32_t node_begin, uint32_t item_begin, const uint2* node_parent, const uint2* item_parent,
                                  const uint32_t* edit_items, const Box* edit_boxes, uint32_t n_edits) {
    const uint32_t ed = blockIdx.x * blockDim.x + threadIdx.x;
    if (ed >= n_edits) return;
    const uint32_t item = edit_items[ed] - item_begin;
    R.item_box[item] = edit_boxes[ed]; R.item_edited[item] = 1u;
    uint32_t n = item_parent[item].x;
    while (n != MAX_UINT) {
        if (atomicExch(&R.dirty[n - node_begin], 1u) != 0u) break;          // someone else marked it and everything above
        const uint32_t up = node_parent[n - node_begin].x;
        if (up != MAX_UINT) atomicAdd(&R.pending[up - node_begin], 1u);
        n = up;
    }
}

__device__ __forceinline__ void refit_node(Node8* nodes, uint32_t n, const RefitState& R, uint32_t node_begin, uint32_t item_begin) {
    Node8 nd = nodes[n];
    const uint32_t used = (uint32_t)nd.imask | (uint32_t)nd.lmask;
    float origin[3] = { nd.ox, nd.oy, nd.oz }; uint8_t e[3] = { nd.ex, nd.ey, nd.ez };
    Box cb[8]; uint32_t changed = 0; bool fits = true;
    uint32_t ci = 0, li = 0;
    for (uint32_t s = 0; s < 8; s++) {
        if ((nd.imask >> s) & 1u) {
            const uint32_t c = nd.child_base + ci++;
            if (R.dirty[c - node_begin]) { cb[s] = R.node_bounds[c - node_begin]; changed |= 1u << s; } else cb[s] = dequantise_child(nd, s);
        } else if ((nd.lmask >> s) & 1u) {
            const uint32_t it = nd.item_base + li++ - item_begin;
            if (R.item_edited[it]) { cb[s] = R.item_box[it]; changed |= 1u << s; } else cb[s] = dequantise_child(nd, s);
        }
    }
    for (uint32_t s = 0; s < 8; s++) if ((changed >> s) & 1u) {
        uint8_t ql[3], qh[3];
        if (quantise_child(origin, e, cb[s], ql, qh)) { for (int k = 0; k < 3; k++) { nd.qlo[k][s] = ql[k]; nd.qhi[k][s] = qh[k]; } }
        else fits = false;
    }
    if (!fits) {   // a child left the node's grid: a new grid over the union of all children (the unchanged ones from their quantised boxes: conservative)
        Box u;
        for (int k = 0; k < 3; k++) { u.lo[k] = 3.0e38f; u.hi[k] = -3.0e38f; }
        for (uint32_t s = 0; s < 8; s++) if ((used >> s) & 1u) for (int k = 0; k < 3; k++) { u.lo[k] = fminf(u.lo[k], cb[s].lo[k]); u.hi[k] = fmaxf(u.hi[k], cb[s].hi[k]); }
        for (int k = 0; k < 3; k++) e[k] = grid_exponent(u.lo[k], u.hi[k]);   // (the grid k_collapse would choose)
        grid_no_axis_much_finer(e);
        for (int k = 0; k < 3; k++) origin[k] = grid_origin(u.lo[k], e[k]);
        nd.ox = origin[0]; nd.oy = origin[1]; nd.oz = origin[2];
        nd.ex = e[0]; nd.ey = e[1]; nd.ez = e[2];
        for (uint32_t s = 0; s < 8; s++) if ((used >> s) & 1u) {
            uint8_t a[3], b[3];
            if (!quantise_child(origin, e, cb[s], a, b)) for (int k = 0; k < 3; k++) { a[k] = 0; b[k] = 255; }   // (an instance without a finite vertex: everything)
            for (int k = 0; k < 3; k++) { nd.qlo[k][s] = a[k]; nd.qhi[k][s] = b[k]; }
        }
    }
    nodes[n] = nd;
    Box nb;   // what the parent has to bound: the union of this node's children as the traversal sees them
    for (int k = 0; k < 3; k++) { nb.lo[k] = 3.0e38f; nb.hi[k] = -3.0e38f; }
    for (uint32_t s = 0; s < 8; s++) if ((used >> s) & 1u) {
        const Box c = dequantise_child(nd, s);
        for (int k = 0; k < 3; k++) { nb.lo[k] = fminf(nb.lo[k], c.lo[k]); nb.hi[k] = fmaxf(nb.hi[k], c.hi[k]); }
    }
    R.node_bounds[n - node_begin] = nb;
}

__global__ void k_tlas_refit_apply(Node8* nodes, RefitState R, uint32_t node_begin, uint32_t item_begin, const uint2* node_parent, const uint2* item_parent,
                                   const uint32_t* edit_items, uint32_t n_edits) {
    const uint32_t ed = blockIdx.x * blockDim.x + threadIdx.x;
    if (ed >= n_edits) return;
    uint32_t n = item_parent[edit_items[ed] - item_begin].x;
    while (n != MAX_UINT) {
        if (atomicAdd(&R.pending[n - node_begin], 0u) != 0u) break;           // a dirty child is still out: whoever finishes the last one comes back here
        if (atomicExch(&R.claimed[n - node_begin], 1u) != 0u) break;          // another edit under this node got here first
        __threadfence();
        refit_node(nodes, n, R, node_begin, item_begin);
        __threadfence();
        const uint32_t up = node_parent[n - node_begin].x;
        if (up == MAX_UINT) break;
        if (atomicSub(&R.pending[up - node_begin], 1u) != 1u) break;          // not the last dirty child of the parent
        n = up;
    }
}

// the traversal's record of every TLAS leaf (msne_device.h TlasLeaf), from the instance records: after a TLAS build and after every in-place update
__global__ void k_tlas_leaves(const uint32_t* items, const InstanceRec* instances, const float4* spheres, uint32_t n, TlasLeaf* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ii = items[i];
    const InstanceRec r = instances[ii];
    TlasLeaf l;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 4; b++) l.w2i[4 * a + b] = r.world_to_instance.m[a][b];
    l.root = (r.flags & INST_FLAG_VISIBLE) ? r.blas_root : MAX_UINT;
    l.inst = (r.flags & INST_FLAG_WORLD) ? WORLD_INSTANCE : ii;
    l.flags = r.flags; l.pad = 0u;
    const float4 sp = spheres[ii];
    l.sx = sp.x; l.sy = sp.y; l.sz = sp.z; l.sr = sp.w;
    out[i] = l;
}
void bvh_tlas_leaves(hipStream_t s, const uint32_t* items, const InstanceRec* instances, const float4* spheres, uint32_t n, TlasLeaf* out) {
    if (n) hipLaunchKernelGGL(k_tlas_leaves, dim3((n + 255) / 256), dim3(256), 0, s, items, instances, spheres, n, out);
}

void bvh_tlas_links(hipStream_t s, const Node8* nodes, uint32_t node_begin, uint32_t node_end, uint32_t item_begin, uint2* node_parent, uint2* item_parent, uint32_t root) {
    if (node_end > node_begin) hipLaunchKernelGGL(k_tlas_links, dim3((node_end - node_begin + 255) / 256), dim3(256), 0, s, nodes, node_begin, node_end, item_begin, node_parent, item_parent, root);
}

// insts / meshes describe the n_edits edited instances (their NEW transforms), edit_items their TLAS leaf items; n_nodes / n_items: the TLAS's nodes and leaf items
bool bvh_refit_tlas(BuildScratch* scratch, hipStream_t s, const TlasInst* insts, const TlasMesh* meshes, uint32_t nmeshes, const uint32_t* edit_items, uint32_t n_edits,
                    Node8* nodes, uint32_t node_begin, uint32_t n_nodes, uint32_t item_begin, uint32_t n_items, const uint2* node_parent, const uint2* item_parent,
                    const uint32_t* edit_ids /* the edited instances, in the order of insts */, float4* spheres /* per instance */) {
    if (n_edits == 0) return true;
    // one buffer of the scratch, carved (256-B aligned pieces) and grown on demand
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t zeroed = ((size_t)3 * n_nodes + n_items) * 4, state_bytes = zeroed + ((size_t)n_nodes + n_items) * sizeof(Box);
    const size_t o_inst = 0, o_boxes = o_inst + up((size_t)n_edits * sizeof(TlasInst)), o_mesh = o_boxes + up((size_t)n_edits * sizeof(Box)),
                 o_items = o_mesh + up((size_t)std::max(nmeshes, 1u) * sizeof(TlasMesh)), o_ids = o_items + up((size_t)n_edits * 4), o_state = o_ids + up((size_t)n_edits * 4), total = o_state + up(state_bytes);
    if (total > scratch->refit_bytes) {
        if (scratch->refit) (void)hipFree(scratch->refit);
        scratch->refit = nullptr; scratch->refit_bytes = 0;
        HIPCHK(hipMalloc(&scratch->refit, total + total / 2));
        scratch->refit_bytes = total + total / 2;
    }
    char* const base = (char*)scratch->refit;
    TlasInst* dinst = (TlasInst*)(base + o_inst); Box* dboxes = (Box*)(base + o_boxes); TlasMesh* dmesh = (TlasMesh*)(base + o_mesh); uint32_t* ditems = (uint32_t*)(base + o_items);
    uint32_t* dids = (uint32_t*)(base + o_ids);
    char* state = base + o_state;
    HIPCHK(hipMemsetAsync(state, 0, zeroed, s));
    RefitState R;
    R.dirty = (uint32_t*)state; R.pending = R.dirty + n_nodes; R.claimed = R.pending + n_nodes; R.item_edited = R.claimed + n_nodes;
    R.node_bounds = (Box*)(state + zeroed); R.item_box = R.node_bounds + n_nodes;
    HIPCHK(hipMemcpyAsync(dinst, insts, (size_t)n_edits * sizeof(TlasInst), hipMemcpyHostToDevice, s));
    if (nmeshes) HIPCHK(hipMemcpyAsync(dmesh, meshes, (size_t)nmeshes * sizeof(TlasMesh), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ditems, edit_items, (size_t)n_edits * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dids, edit_ids, (size_t)n_edits * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_instance_boxes, dim3(n_edits), dim3(256), 0, s, dinst, dmesh, n_edits, dboxes, dids, spheres);
    hipLaunchKernelGGL(k_tlas_refit_mark, dim3((n_edits + 63) / 64), dim3(64), 0, s, R, node_begin, item_begin, node_parent, item_parent, ditems, dboxes, n_edits);
    hipLaunchKernelGGL(k_tlas_refit_apply, dim3((n_edits + 63) / 64), dim3(64), 0, s, nodes, R, node_begin, item_begin, node_parent, item_parent, ditems, n_edits);
    HIPCHK(hipStreamSynchronize(s));
    return true;
}


}  // namespace msne
