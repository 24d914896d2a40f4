// trace.hip — k_trace_closest / k_trace_shadow: software replacement for the reference's
// TraceRay calls (shaders/hrtsystem/intersection.hlsl:18-22 closest hit, :33-46 any hit; payload
// written by closesthit/miss/shadowmiss in main.hlsl:102-118) over the driver-built TLAS/BLAS
// (engine/hrtsystem/Accel.zig:94-184,484).
//
// Design (gfx950): persistent waves.  A wave reserves a private chunk of the compacted ray queue with one
// atomicAdd (its first chunk is static — no atomic at all), and every LANE refills itself from that chunk
// the moment its ray terminates (ballot + prefix popcount), so the wave never idles on its slowest ray.
// Each lane walks the two-level 8-wide quantized BVH (80-B nodes, 48-B triangles) with a short per-lane
// stack in LDS ([entry][thread] layout → conflict-free ds_read/ds_write_b32), spilling to HBM only beyond
// STACK_LDS entries.  No MFMA: this is pointer chasing + 3-vector math.
// Box tests use fmaf and a relative slack (they only gate which triangles are tested); the triangle test is
// the watertight Woop–Benthin–Wald test evaluated op-for-op like the test oracle, and equal-t ties resolve
// to the smallest (instance, geometry, primitive), so results do not depend on BVH shape or visit order.
#include "msne_device.h"

namespace msne {

constexpr int TRACE_BLOCK = 256;
#ifndef TRACE_WPS
#define TRACE_WPS 6          // resident waves per SIMD the trace kernels are register-allocated for (= blocks of 256 per CU)
#endif
constexpr int STACK_LDS = 24;
constexpr int STACK_SPILL = 232;          // total depth 256 entries per lane
constexpr uint32_t ENT_SENTINEL = 0xFFFFFFFFu;     // leaving a transformed instance: restore the world-space ray
constexpr uint32_t ENT_SENTINEL_ID = 0xFFFFFFFEu;  // leaving an identity-transform instance
constexpr uint32_t ENT_KIND_TRI = 1u << 30;
constexpr uint32_t ENT_KIND_INST = 2u << 30;
struct TraceTune { uint32_t refill, t_node, t_tri, t_inst; };   // lane-refill and phase-vote thresholds (lanes of 64)

struct RayK { int kx, ky, kz; float Sx, Sy, Sz; };

__device__ __forceinline__ float idx3(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

__device__ __forceinline__ RayK rayk_make(f3 d) {
    RayK k;
    k.kz = 0;
    if (absf(d.y) > absf(d.x)) k.kz = 1;
    if (absf(d.z) > absf(idx3(d, k.kz))) k.kz = 2;
    k.kx = k.kz + 1; if (k.kx == 3) k.kx = 0;
    k.ky = k.kx + 1; if (k.ky == 3) k.ky = 0;
    if (idx3(d, k.kz) < 0.0f) { int t = k.kx; k.kx = k.ky; k.ky = t; }
    float dz = idx3(d, k.kz);
    k.Sx = idx3(d, k.kx) / dz; k.Sy = idx3(d, k.ky) / dz; k.Sz = 1.0f / dz;
    return k;
}

// watertight ray/triangle test (Woop, Benthin, Wald 2013), no culling; (u,v) = weights of vertices 1,2.
// Written without early-outs (one predicate at the end) so a wave does not fragment into exec-mask branches;
// the arithmetic is exactly the test oracle's.
__device__ __forceinline__ bool tri_intersect(f3 o, const RayK& k, f3 v0, f3 v1, f3 v2, float& t, float& u, float& v) {
    const f3 A = sub(v0, o), B = sub(v1, o), C = sub(v2, o);
    const float Akz = idx3(A, k.kz), Bkz = idx3(B, k.kz), Ckz = idx3(C, k.kz);
    const float Ax = idx3(A, k.kx) - k.Sx * Akz, Ay = idx3(A, k.ky) - k.Sy * Akz;
    const float Bx = idx3(B, k.kx) - k.Sx * Bkz, By = idx3(B, k.ky) - k.Sy * Bkz;
    const float Cx = idx3(C, k.kx) - k.Sx * Ckz, Cy = idx3(C, k.ky) - k.Sy * Ckz;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (__builtin_expect(U == 0.0f || V == 0.0f || W == 0.0f, 0)) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx; U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx; V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax; W = (float)(BxAy - ByAx);
    }
    const bool mixed = (U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f);
    const float det = U + V + W;
    const float Az = k.Sz * Akz, Bz = k.Sz * Bkz, Cz = k.Sz * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const float rcp = 1.0f / det;
    const float tt = T * rcp;
    t = tt; u = V * rcp; v = W * rcp;
    return !mixed && det != 0.0f && tt > 0.0f;
}

__device__ __forceinline__ float safe_inv(float d) { return absf(d) < 1e-30f ? (d < 0.0f ? -1e30f : 1e30f) : 1.0f / d; }

struct Hit { uint32_t inst, geo, prim, tri; float t, u, v; };   // tri = slot of the hit triangle's record in SceneView::tris

// per-lane traversal state
struct Lane {
    f3 o_w, d_w;      // world-space ray
    f3 o, d, id;      // current-space ray and reciprocal direction
    RayK rk;
    Hit best;
    uint32_t cur;     // entry being processed when `have`
    uint32_t cur_inst;
    int sp;
    bool have, in_blas;
};

struct StackRef { uint32_t* lds; uint32_t* spill; uint32_t spill_stride; uint32_t* overflow; };

__device__ __forceinline__ void lane_push(Lane& L, const StackRef& S, uint32_t e) {   // general (slow-path) push
    if (L.sp < STACK_LDS) S.lds[L.sp * TRACE_BLOCK + threadIdx.x] = e;
    else if (L.sp < STACK_LDS + STACK_SPILL) S.spill[(size_t)(L.sp - STACK_LDS) * S.spill_stride] = e;
    else { *S.overflow = 1u; return; }
    L.sp++;
}
__device__ __forceinline__ uint32_t lane_pop(Lane& L, const StackRef& S) {
    L.sp--;
    if (__builtin_expect(L.sp >= STACK_LDS, 0)) return S.spill[(size_t)(L.sp - STACK_LDS) * S.spill_stride];
    return S.lds[L.sp * TRACE_BLOCK + threadIdx.x];
}

__device__ __forceinline__ void lane_set_space(Lane& L, f3 o, f3 d) {
    L.o = o; L.d = d;
    L.id = F3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    L.rk = rayk_make(d);
}

__device__ __forceinline__ void lane_begin(Lane& L, const SceneView& sc, f3 o, f3 d, float tmax) {
    L.o_w = o; L.d_w = d;
    lane_set_space(L, o, d);
    L.best.inst = MAX_UINT; L.best.geo = 0; L.best.prim = 0; L.best.tri = 0; L.best.t = tmax; L.best.u = 0.0f; L.best.v = 0.0f;
    L.sp = 0; L.in_blas = sc.root_in_blas != 0u; L.cur_inst = WORLD_INSTANCE;   // only used when the root IS the world BLAS
    L.cur = sc.tlas_root; L.have = sc.tlas_root != MAX_UINT;
}

// ---- the three step bodies.  The wave loop decides, per iteration, which bodies run (see trace_wave_loop). ----

// internal node: 5 x 16-B loads, 8 quantised box tests; the nearest hit child becomes the current entry, the others are
// pushed.  Straight-line code: every child slot computes its entry and issues an UNCONDITIONAL ds_write — to its stack
// position if it was hit, to a per-thread trash row otherwise — so the hot path has no exec-mask branches at all
// (the earlier per-child conditional pushes cost ~76 branches / ~310 SALU instructions per node visit).
template <bool STATS>
__device__ __forceinline__ void step_node(Lane& L, const SceneView& sc, const StackRef& S, unsigned long long& nv) {
    const uint32_t cur = L.cur;
    const uint4* np = reinterpret_cast<const uint4*>(sc.nodes + cur);
    const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3], w4 = np[4];
    if (STATS) nv++;
    const float nox = u2f(w0.x), noy = u2f(w0.y), noz = u2f(w0.z);
    const uint32_t ex = w0.w & 0xff, ey = (w0.w >> 8) & 0xff, ez = (w0.w >> 16) & 0xff, imask = w0.w >> 24;
    const uint32_t child_base = w1.x, item_base = w1.y;
    const uint32_t meta_lo = w1.z, meta_hi = w1.w;
    const float ax = u2f(ex << 23) * L.id.x, ay = u2f(ey << 23) * L.id.y, az = u2f(ez << 23) * L.id.z;
    const float bx = (nox - L.o.x) * L.id.x, by = (noy - L.o.y) * L.id.y, bz = (noz - L.o.z) * L.id.z;
    // byte planes: qlo[0] = w2.xy, qlo[1] = w2.zw, qlo[2] = w3.xy, qhi[0] = w3.zw, qhi[1] = w4.xy, qhi[2] = w4.zw
    const uint32_t qw[12] = { w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, w4.z, w4.w };
    const float tlimit = L.best.t;
    const uint32_t leaf_kind = L.in_blas ? ENT_KIND_TRI : ENT_KIND_INST;
    const uint32_t leaf_cnt_mask = L.in_blas ? 3u : 0u;
    uint32_t ent[8]; uint32_t hitmask = 0; uint32_t next_entry = 0; float nearest_t = 3.0e38f; uint32_t nearest_bit = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int wi = i >> 2, sh = (i & 3) * 8;
        const float lx = (float)((qw[0 + wi] >> sh) & 0xff), ly = (float)((qw[2 + wi] >> sh) & 0xff), lz = (float)((qw[4 + wi] >> sh) & 0xff);
        const float hx = (float)((qw[6 + wi] >> sh) & 0xff), hy = (float)((qw[8 + wi] >> sh) & 0xff), hz = (float)((qw[10 + wi] >> sh) & 0xff);
        const float t0x = __builtin_fmaf(lx, ax, bx), t1x = __builtin_fmaf(hx, ax, bx);
        const float t0y = __builtin_fmaf(ly, ay, by), t1y = __builtin_fmaf(hy, ay, by);
        const float t0z = __builtin_fmaf(lz, az, bz), t1z = __builtin_fmaf(hz, az, bz);
        const float n = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
        const float f = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tlimit));
        const uint32_t m = ((i < 4 ? meta_lo : meta_hi) >> sh) & 0xff;
        const bool internal = (imask >> i) & 1u;
        const bool hit = (internal || m != 0xffu) && n <= f * 1.00001f;
        const uint32_t e_int = child_base + __popc(imask & ((1u << i) - 1u));
        const uint32_t e_leaf = leaf_kind | (((m >> 5) & leaf_cnt_mask) << 28) | (item_base + (m & 31u));
        ent[i] = internal ? e_int : e_leaf;
        hitmask |= hit ? (1u << i) : 0u;
        const bool nearer = hit && n < nearest_t;
        nearest_t = nearer ? n : nearest_t;
        next_entry = nearer ? ent[i] : next_entry;
        nearest_bit = nearer ? (1u << i) : nearest_bit;
    }
    const uint32_t rest = hitmask & ~nearest_bit;
    const uint32_t npush = __popc(rest);
    if (__builtin_expect(L.sp + (int)npush > STACK_LDS, 0)) {
        // rare deep path: general pushes (LDS, then the HBM spill area)
#pragma unroll
        for (int i = 0; i < 8; i++) if ((rest >> i) & 1u) lane_push(L, S, ent[i]);
    } else {
        uint32_t* row0 = S.lds + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t pos = ((rest >> i) & 1u) ? (uint32_t)L.sp + __popc(rest & ((1u << i) - 1u)) : (uint32_t)STACK_LDS;   // STACK_LDS = trash row
            row0[pos * TRACE_BLOCK] = ent[i];
        }
        L.sp += (int)npush;
    }
    L.cur = next_entry; L.have = hitmask != 0u;
}

// ONE triangle of a leaf per call (entry = kind | (remaining-1) << 28 | index), so lanes stay in step whatever the
// leaf sizes.  Returns true when an any-hit ray is finished.
template <bool ANY_HIT, bool STATS>
__device__ __forceinline__ bool step_tri(Lane& L, const SceneView& sc, unsigned long long& nt) {
    const uint32_t cur = L.cur;
    const uint32_t rem = (cur >> 28) & 3u, idx = cur & 0x0FFFFFFFu;
    if (rem) L.cur = ENT_KIND_TRI | ((rem - 1u) << 28) | (idx + 1u); else L.have = false;
    const uint4* tp = reinterpret_cast<const uint4*>(sc.tris + idx);
    const uint4 a = tp[0], b = tp[1], c = tp[2];
    if (STATS) nt++;
    float t, u, v;
    const bool hit = tri_intersect(L.o, L.rk, F3(u2f(a.x), u2f(a.y), u2f(a.z)), F3(u2f(a.w), u2f(b.x), u2f(b.y)), F3(u2f(b.z), u2f(b.w), u2f(c.x)), t, u, v);
    const uint32_t inst = L.cur_inst == WORLD_INSTANCE ? c.w : L.cur_inst;   // world BLAS: the triangle record names its instance
    if (ANY_HIT) {
        const bool occluded = hit && t < L.best.t;
        L.best.inst = occluded ? inst : L.best.inst;
        return occluded;
    }
    const bool tie = t == L.best.t && L.best.inst != MAX_UINT
        && (inst < L.best.inst || (inst == L.best.inst && (c.y < L.best.geo || (c.y == L.best.geo && c.z < L.best.prim))));
    const bool closer = hit && (t < L.best.t || tie);
    L.best.t = closer ? t : L.best.t; L.best.u = closer ? u : L.best.u; L.best.v = closer ? v : L.best.v;
    L.best.inst = closer ? inst : L.best.inst; L.best.geo = closer ? c.y : L.best.geo; L.best.prim = closer ? c.z : L.best.prim;
    L.best.tri = closer ? idx : L.best.tri;
    return false;
}

// instance leaf (enter the BLAS in instance space; t is preserved: d is not renormalised) or sentinel (leave it)
__device__ __forceinline__ void step_inst(Lane& L, const SceneView& sc, const StackRef& S) {
    const uint32_t cur = L.cur;
    L.have = false;
    if (cur >= ENT_SENTINEL_ID) {
        if (cur == ENT_SENTINEL) lane_set_space(L, L.o_w, L.d_w);
        L.in_blas = false;
        return;
    }
    const uint32_t item = cur & 0x3FFFFFFFu;
    const uint32_t ii = sc.tlas_items[item];
    const InstanceRec* ir = sc.instances + ii;
    const uint32_t root = ir->blas_root, flags = ir->flags;
    if (!(flags & 1u) || root == MAX_UINT) return;
    if (flags & 2u) {
        lane_push(L, S, ENT_SENTINEL_ID);   // identity transform: M·(o,1) = o and M·d = d exactly, the ray is left as is
    } else {
        const float4* mp = reinterpret_cast<const float4*>(&ir->world_to_instance);
        const float4 r0 = mp[0], r1 = mp[1], r2 = mp[2];
        m34 M;
        M.m[0][0] = r0.x; M.m[0][1] = r0.y; M.m[0][2] = r0.z; M.m[0][3] = r0.w;
        M.m[1][0] = r1.x; M.m[1][1] = r1.y; M.m[1][2] = r1.z; M.m[1][3] = r1.w;
        M.m[2][0] = r2.x; M.m[2][1] = r2.y; M.m[2][2] = r2.z; M.m[2][3] = r2.w;
        lane_set_space(L, m34_mul_point(M, L.o_w), m34_mul_vec(M, L.d_w));
        lane_push(L, S, ENT_SENTINEL);
    }
    L.in_blas = true; L.cur_inst = (flags & INST_FLAG_WORLD) ? WORLD_INSTANCE : ii;
    L.cur = root; L.have = true;
}

// ---------------------------------------------------------------------------------------------
// Persistent-wave dequeue.  One device-scope atomic word saturates at ~88 dequeues/us on MI355X
// (MI355X_MICROARCH.md "dequeue"), so (1) every wave's FIRST chunk is static (wave w takes chunk w: an
// empty or short queue costs no atomics at all), (2) later chunks come from one atomicAdd per wave on the
// queue head, and (3) a chunk is 64..512 rays that the wave's lanes consume one by one as they go idle.
struct WaveQueue {
    uint32_t n, chunk, pos, end, nwaves_chunk;
    uint32_t* head;
    bool exhausted;
    __device__ WaveQueue(uint32_t n_, uint32_t* head_) : n(n_), head(head_), exhausted(false) {
        const uint32_t nwaves = gridDim.x * (TRACE_BLOCK / 64);
        uint32_t c = (n / (nwaves * 4u) + 63u) & ~63u;
        chunk = c < 64u ? 64u : (c > 512u ? 512u : c);
        nwaves_chunk = nwaves * chunk;
        const uint32_t wave = blockIdx.x * (TRACE_BLOCK / 64) + (threadIdx.x >> 6);
        pos = wave * chunk; end = pos + chunk;
        if (end > n) end = n;
        if (pos >= n) { pos = end = n; exhausted = true; }   // later (atomic) chunks lie beyond every static chunk
    }
    // hands out up to `want` consecutive ray indices starting at the returned base (wave-uniform); 0 when the queue is exhausted
    __device__ uint32_t take(uint32_t want, uint32_t& base) {
        if (pos >= end && !exhausted) {
            uint32_t b = 0;
            if ((threadIdx.x & 63u) == 0) b = atomicAdd(head, chunk);
            b = __builtin_amdgcn_readfirstlane(b);
            pos = nwaves_chunk + b; end = pos + chunk;
            if (end > n) end = n;
            if (pos >= n) { pos = end = n; exhausted = true; }
        }
        const uint32_t avail = end - pos;
        const uint32_t got = want < avail ? want : avail;
        base = pos; pos += got;
        return got;
    }
};

// Wave loop shared by the three kernels.  `load(i, o, d, tmax)` returns false for entries without a ray;
// `store(i, lane)` receives the finished lane.
template <bool ANY_HIT, bool STATS, class Load, class Store>
__device__ __forceinline__ void trace_wave_loop(const SceneView& sc, uint32_t n, uint32_t* head, uint32_t* lds_stack, uint32_t* spill, uint32_t* overflow,
                                                TraceTune tune, Load load, Store store, unsigned long long& nv, unsigned long long& nt, unsigned long long* prof) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t gtid = blockIdx.x * TRACE_BLOCK + threadIdx.x;
    StackRef S{ lds_stack, spill + gtid, gridDim.x * TRACE_BLOCK, overflow };
    WaveQueue wq(n, head);
    Lane L; L.have = false; L.sp = 0;
    bool active = false; uint32_t my = 0;
    // STATS builds: wave-cycle profile of the loop sections (s_memtime), accumulated per wave
    unsigned long long cyc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tprev = 0;
    auto lap = [&](int k) { if (STATS) { const unsigned long long t = __builtin_readcyclecounter(); cyc[k] += t - tprev; tprev = t; } };
    if (STATS) tprev = __builtin_readcyclecounter();
    for (;;) {
        if (STATS) cyc[7] += 1;   // iterations
        // (a) lanes without a current entry pop one, or finish when their stack is empty
        if (active && !L.have) {
            if (L.sp == 0) { store(my, L); active = false; }
            else { L.cur = lane_pop(L, S); L.have = true; }
        }
        lap(0);
        // (b) idle lanes take new rays from the wave's private chunk
        const unsigned long long act = __ballot(active);
        const uint32_t nidle = 64u - (uint32_t)__popcll(act);   // blocks are 4 full waves
        if (!wq.exhausted && nidle >= tune.refill) {
            uint32_t base;
            const uint32_t got = wq.take(nidle, base);
            const uint32_t r = (uint32_t)__popcll(~act & lt);
            if (!active && r < got) {
                my = base + r;
                f3 o, d; float tmax;
                if (load(my, o, d, tmax)) { lane_begin(L, sc, o, d, tmax); active = L.have; if (!active) store(my, L); }
                else { L.best.inst = MAX_UINT; L.best.geo = 0; L.best.prim = 0; L.best.tri = 0; L.best.u = 0.0f; L.best.v = 0.0f; L.best.t = 0.0f; store(my, L); }
            }
        }
        lap(1);
        if (!__ballot(active)) { if (wq.exhausted) break; continue; }
        // (c) phase vote: an expensive body runs only when enough lanes want it (or it is the most wanted one), so
        // lanes of the same kind are batched over time instead of every body running at low utilisation every iteration
        const uint32_t kind = !active ? 4u : (L.cur >= ENT_SENTINEL_ID ? 2u : (L.cur >> 30));
        const uint32_t nn = (uint32_t)__popcll(__ballot(kind == 0u)), nt_ = (uint32_t)__popcll(__ballot(kind == 1u)), ni = (uint32_t)__popcll(__ballot(kind == 2u));
        const bool do_n = nn && (nn >= tune.t_node || (nn >= nt_ && nn >= ni));
        const bool do_t = nt_ && (nt_ >= tune.t_tri || (nt_ > nn && nt_ >= ni));
        const bool do_i = ni && (ni >= tune.t_inst || (ni > nn && ni > nt_));
        lap(2);
        if (do_n && kind == 0u) step_node<STATS>(L, sc, S, nv);
        lap(3);
        if (STATS && do_n) cyc[6] += __popcll(__ballot(kind == 0u));   // node-lane steps
        if (do_t && kind == 1u) { if (step_tri<ANY_HIT, STATS>(L, sc, nt)) { L.sp = 0; L.have = false; store(my, L); active = false; } }
        lap(4);
        if (do_i && kind == 2u) step_inst(L, sc, S);
        lap(5);
    }
    if (STATS && (threadIdx.x & 63u) == 0) for (int k = 0; k < 8; k++) atomicAdd(&prof[k], cyc[k]);
}

template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK, TRACE_WPS) void k_trace_closest(SceneView sc, PathState st, HitBuf hits, Counters* cnt,
                                                                uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out, TraceTune tune) {
    __shared__ uint32_t lds_stack[(STACK_LDS + 1) * TRACE_BLOCK];   // + one trash row for the branch-free pushes
    const uint32_t n = cnt->n_cur;
    unsigned long long nv = 0, nt = 0;
    trace_wave_loop<false, STATS>(sc, n, &cnt->head_closest, lds_stack, spill, overflow, tune,
        [&](uint32_t i, f3& o, f3& d, float& tmax) -> bool {
            const float4 ro = st.ro[i];
            if (f2u(ro.w) & PATH_FLAG_ZOMBIE) return false;
            const float4 rd = st.rd[i];
            o = F3(ro.x, ro.y, ro.z); d = F3(rd.x, rd.y, rd.z); tmax = INFINITY_F;
            return true;
        },
        [&](uint32_t i, const Lane& L) {
            hits.rec[i] = make_uint4(L.best.inst, L.best.tri, f2u(L.best.u), f2u(L.best.v));
        }, nv, nt, stat_out + 4);
    if (STATS) { atomicAdd(&stat_out[0], nv); atomicAdd(&stat_out[1], nt); }
}

template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK, TRACE_WPS) void k_trace_shadow(SceneView sc, ShadowQueue q, PathState next, Counters* cnt,
                                                               uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out, TraceTune tune) {
    __shared__ uint32_t lds_stack[(STACK_LDS + 1) * TRACE_BLOCK];   // + one trash row for the branch-free pushes
    const uint32_t n = cnt->n_shadow;
    unsigned long long nv = 0, nt = 0;
    trace_wave_loop<true, STATS>(sc, n, &cnt->head_shadow, lds_stack, spill, overflow, tune,
        [&](uint32_t i, f3& o, f3& d, float& tmax) -> bool {
            const float4 qo = q.o[i], qd = q.d[i];
            o = F3(qo.x, qo.y, qo.z); d = F3(qd.x, qd.y, qd.z); tmax = qo.w;
            return true;
        },
        [&](uint32_t i, const Lane& L) {
            if (L.best.inst != MAX_UINT) {   // ShadowIntersection::hit → lightSample.pdf = 0 (light.hlsl:75-77,154-156): the pending contribution vanishes
                const uint32_t tg = f2u(q.d[i].w), j = tg >> 1;
                if (tg & 1u) { next.p1x[j] = 0.0f; next.p1y[j] = 0.0f; next.p1z[j] = 0.0f; }
                else { next.p0x[j] = 0.0f; next.p0y[j] = 0.0f; next.p0z[j] = 0.0f; }
            }
        }, nv, nt, stat_out + 12);
    if (STATS) { atomicAdd(&stat_out[2], nv); atomicAdd(&stat_out[3], nt); }
}

// probe kernel for tests: arbitrary rays → hit records (closest) or occlusion flags (any)
template <bool ANY_HIT>
__global__ __launch_bounds__(TRACE_BLOCK, TRACE_WPS) void k_trace_probe(SceneView sc, const float* rays /*7 per ray: o,d,tmax*/, uint32_t n, uint32_t* head,
                                                              uint32_t* out_ids /*4 per ray: hit,inst,geo,prim*/, float* out_tuv /*3 per ray*/,
                                                              uint32_t* spill, uint32_t* overflow, TraceTune tune) {
    __shared__ uint32_t lds_stack[(STACK_LDS + 1) * TRACE_BLOCK];   // + one trash row for the branch-free pushes
    unsigned long long nv = 0, nt = 0;
    trace_wave_loop<ANY_HIT, false>(sc, n, head, lds_stack, spill, overflow, tune,
        [&](uint32_t i, f3& o, f3& d, float& tmax) -> bool {
            const float* r = rays + 7 * (size_t)i;
            o = F3(r[0], r[1], r[2]); d = F3(r[3], r[4], r[5]); tmax = r[6];
            return true;
        },
        [&](uint32_t i, const Lane& L) {
            out_ids[4 * i] = L.best.inst != MAX_UINT ? 1u : 0u; out_ids[4 * i + 1] = L.best.inst; out_ids[4 * i + 2] = L.best.geo; out_ids[4 * i + 3] = L.best.prim;
            out_tuv[3 * i] = L.best.t; out_tuv[3 * i + 1] = L.best.u; out_tuv[3 * i + 2] = L.best.v;
        }, nv, nt, nullptr);
}

// ---------------- host launch wrappers ----------------
void launch_trace_closest(hipStream_t s, int grid, bool stats, const SceneView& sc, const PathState& st, const HitBuf& hits, Counters* cnt,
                          uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out, const uint32_t tune4[4]) {
    const TraceTune tune{ tune4[0], tune4[1], tune4[2], tune4[3] };
    if (stats) hipLaunchKernelGGL(k_trace_closest<true>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, st, hits, cnt, spill, overflow, stat_out, tune);
    else hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, st, hits, cnt, spill, overflow, stat_out, tune);
}
void launch_trace_shadow(hipStream_t s, int grid, bool stats, const SceneView& sc, const ShadowQueue& q, const PathState& next, Counters* cnt,
                         uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out, const uint32_t tune4[4]) {
    const TraceTune tune{ tune4[0], tune4[1], tune4[2], tune4[3] };
    if (stats) hipLaunchKernelGGL(k_trace_shadow<true>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, q, next, cnt, spill, overflow, stat_out, tune);
    else hipLaunchKernelGGL(k_trace_shadow<false>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, q, next, cnt, spill, overflow, stat_out, tune);
}
void launch_trace_probe(hipStream_t s, int grid, const SceneView& sc, const float* rays, uint32_t n, int any_hit, uint32_t* head, uint32_t* out_ids, float* out_tuv,
                        uint32_t* spill, uint32_t* overflow, const uint32_t tune4[4]) {
    const TraceTune tune{ tune4[0], tune4[1], tune4[2], tune4[3] };
    if (any_hit) hipLaunchKernelGGL(k_trace_probe<true>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, rays, n, head, out_ids, out_tuv, spill, overflow, tune);
    else hipLaunchKernelGGL(k_trace_probe<false>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, rays, n, head, out_ids, out_tuv, spill, overflow, tune);
}
int trace_blocks_per_cu() { return TRACE_WPS; }
size_t trace_spill_words(int grid) { return (size_t)grid * TRACE_BLOCK * STACK_SPILL; }

}  // namespace msne
