// trace.hip — k_trace_closest / k_trace_shadow: software replacement for the reference's
// TraceRay calls (shaders/hrtsystem/intersection.hlsl:18-22 closest hit, :33-46 any hit; payload
// written by closesthit/miss/shadowmiss in main.hlsl:102-118) over the driver-built TLAS/BLAS
// (engine/hrtsystem/Accel.zig:94-184,484).
//
// Design (gfx950).  The kernels are VALU-issue bound (rocprofv3: SQ_ACTIVE_INST_VALU ≈ 90 % of SIMD time,
// HBM traffic far below peak), so everything here minimises vector instructions per ray and keeps lanes busy:
//  * persistent waves: a wave reserves a private chunk of the compacted ray queue with one atomicAdd (its first
//    chunk is static), and every LANE refills itself from that chunk when its ray terminates;
//  * the per-lane stack (LDS, [entry][word][thread] → conflict-free) holds child GROUPS, not children: one
//    2-word entry {base, hit bits | type bits} names every hit-but-unvisited internal child of a node, so a node
//    visit pushes at most once and never materialises per-child entries;
//  * the builder places a node's children in octant order (slot bit k set = child on the + side of axis k), so
//    the visit order is a 2-KB LDS table lookup lut[ray octant][hit bits] instead of a distance sort;
//  * near/far planes are chosen by ray sign once per node (12 v_cndmask), not by min/max per child;
//  * every leaf holds one triangle; the hit leaves of a node form a triangle group {item base, hit bits | leaf mask}, two
//    of which can wait per lane: node traversal runs ahead of the triangle tests (one per iteration), so both bodies
//    run with more lanes; a node visited with a not yet shortened ray costs a few extra visits, never a wrong result;
//  * entering an instance (reload the world-space direction, transform, shear constants) is a "space body" that a wave runs when
//    SPACE_MIN_LANES lanes wait for it or nothing else can be done; the lane puts the world-space half of its ray (origin, reciprocal
//    direction, octant: 7 registers) aside and takes it back inline when its stack is back at the height of the entry (Lane::ret_sp: no sentinel
//    entry, no second pop) — leaving costs no body and no wait.  Both instantiations run 6 waves per SIMD (80 registers; the two-level one spills 13 of them and is still
//    3 % faster than at 5 waves: TRACE_WPS_TLAS below); scenes without a TLAS level run an instantiation without any of the space bookkeeping;
//  * a triangle is tested from a 64-B record that holds every vertex as x y z x y (msne_device.h TriRot): the watertight test's permutation of the vertices by the ray's dominant
//    axis is the address of three 12-B loads, not 18 selects per test (profiles/r05_tri_density.txt).
// Box tests use fmaf and a relative slack (they only gate which triangles are tested); the triangle test is
// the watertight Woop–Benthin–Wald test evaluated op-for-op like the test oracle, and equal-t ties resolve
// to the smallest (instance, geometry, primitive), so results do not depend on BVH shape or visit order.
#include "msne_device.h"

namespace msne {

constexpr int TRACE_BLOCK = 256;
#ifndef TRACE_WPS
#define TRACE_WPS 6          // resident waves per SIMD the trace kernels are register-allocated for (= blocks of 256 per CU)
#endif
// Two-level scenes: the world-space half of the ray a lane keeps while inside an instance costs 13 spilled registers at 80 —
// and 6 waves per SIMD are still 3 % faster than 5 at 96 registers without spills (S2 3597 / 3481 Mrays/s,
// profiles/r04_s2_lane_use.txt "6 instead of 5 waves per SIMD"; round 3 measured the opposite before the leaf records took
// the InstanceRec hop out of the space body).  7 and 8 waves per SIMD — 72 / 64 registers, LDS stacks of 10 / 9 entries —
// lose 2-3 % / 12-20 % (profiles/r05_tri_density.txt).  With both at 6 the grid rescaling in the launch wrappers is the identity.
#ifndef TRACE_WPS_TLAS
#define TRACE_WPS_TLAS 6
#endif
#ifndef TRACE_TRI_MIN_LANES
#define TRACE_TRI_MIN_LANES 1
#endif
#ifndef TRACE_KEEP_RO_TLAS
#define TRACE_KEEP_RO_TLAS 0   // two-level scenes: the rotated origin of the triangle test lives in registers (1) or is rotated again at every test (0)
#endif
#ifndef TRACE_SPACE_MIN_LANES
#define TRACE_SPACE_MIN_LANES 16   // lanes that must wait for a change of space (instance entry / exit) before the wave runs that body
#endif
#ifndef TRACE_STACK_LDS
#define TRACE_STACK_LDS 12
#endif
constexpr uint32_t SPACE_MIN_LANES = TRACE_SPACE_MIN_LANES;
constexpr int STACK_LDS = TRACE_STACK_LDS;   // group entries per lane kept in LDS (2 words each); 6 x (24 KB + 2 KB table) fit the CU's 160 KB
constexpr int STACK_SPILL = 128 - STACK_LDS; // further entries per lane in HBM (2 words each)
constexpr uint32_t GRP_NODE = 0u, GRP_INST = 1u << 16, GRP_KIND_MASK = 3u << 16;

// The ray's half of the watertight test.  kz = the dominant axis of the direction.  Woop et al. (and the test oracle) take kx = kz + 1, ky = kz + 2 (mod 3) and swap the two
// when d[kz] < 0.  Here the permutation is always the ROTATION by kz — the triangle records serve it by address (msne_device.h TriRot) — and the swap is applied where it
// matters: with P / Q = the sheared coordinates along kz + 1 / kz + 2, the oracle's (x, y) are (P, Q) without the swap and (Q, P) with it, so its U = Cx*By - Cy*Bx is
// p - q or q - p of the SAME two products (multiplication commutes exactly), likewise V and W: one select of the operand order per edge function reproduces the oracle's
// bits, signed zeros included (p - q and q - p are exact negations of each other unless the result is zero, and zero results are what edge and vertex hits are made of).
// S1 = d[kz + 1] / d[kz], S2 = d[kz + 2] / d[kz] (the oracle's Sx, Sy in one order or the other), Sz = 1 / d[kz]; ro = the origin rotated like the vertices.
struct RayK { uint32_t kz; float S1, S2, Sz; f3 ro; };

__device__ __forceinline__ float idx3(f3 v, uint32_t i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
__device__ __forceinline__ f3 rot3(f3 v, uint32_t kz) {   // (v[kz], v[kz + 1], v[kz + 2])
    return F3(kz == 0 ? v.x : (kz == 1 ? v.y : v.z), kz == 0 ? v.y : (kz == 1 ? v.z : v.x), kz == 0 ? v.z : (kz == 1 ? v.x : v.y));
}

__device__ __forceinline__ RayK rayk_make(f3 o, f3 d) {
    RayK k;
    k.kz = 0;
    if (absf(d.y) > absf(d.x)) k.kz = 1;
    if (absf(d.z) > absf(idx3(d, k.kz))) k.kz = 2;
    const f3 dr = rot3(d, k.kz);   // (d[kz], d[kz + 1], d[kz + 2])
    k.S1 = dr.y / dr.x; k.S2 = dr.z / dr.x; k.Sz = 1.0f / dr.x;
    k.ro = rot3(o, k.kz);
    return k;
}

// watertight ray/triangle test (Woop, Benthin, Wald 2013), no culling; (u,v) = weights of vertices 1,2.  The vertices arrive rotated like the ray: (v[kz], v[kz + 1], v[kz + 2]).
// Written without early-outs (one predicate at the end) so a wave does not fragment into exec-mask branches;
// the arithmetic is the test oracle's, operation for operation (see RayK for the swap).
__device__ __forceinline__ bool tri_intersect(const RayK& k, f3 r0, f3 r1, f3 r2, float& t, float& u, float& v) {
    const float Akz = r0.x - k.ro.x, Ak1 = r0.y - k.ro.y, Ak2 = r0.z - k.ro.z;
    const float Bkz = r1.x - k.ro.x, Bk1 = r1.y - k.ro.y, Bk2 = r1.z - k.ro.z;
    const float Ckz = r2.x - k.ro.x, Ck1 = r2.y - k.ro.y, Ck2 = r2.z - k.ro.z;
    const float AP = Ak1 - k.S1 * Akz, AQ = Ak2 - k.S2 * Akz;
    const float BP = Bk1 - k.S1 * Bkz, BQ = Bk2 - k.S2 * Bkz;
    const float CP = Ck1 - k.S1 * Ckz, CQ = Ck2 - k.S2 * Ckz;
    const bool swp = k.Sz < 0.0f;   // d[kz] < 0: the oracle's (x, y) = (Q, P)
    const float pu = CP * BQ, qu = CQ * BP, pv = AP * CQ, qv = AQ * CP, pw = BP * AQ, qw = BQ * AP;
    float U = swp ? qu - pu : pu - qu, V = swp ? qv - pv : pv - qv, W = swp ? qw - pw : pw - qw;
    if (__builtin_expect(U == 0.0f || V == 0.0f || W == 0.0f, 0)) {
        const double dpu = (double)CP * (double)BQ, dqu = (double)CQ * (double)BP; U = (float)(swp ? dqu - dpu : dpu - dqu);
        const double dpv = (double)AP * (double)CQ, dqv = (double)AQ * (double)CP; V = (float)(swp ? dqv - dpv : dpv - dqv);
        const double dpw = (double)BP * (double)AQ, dqw = (double)BQ * (double)AP; W = (float)(swp ? dqw - dpw : dpw - dqw);
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

// reciprocal direction for the box tests only (they carry a 1e-5 slack): v_rcp_f32, 1 ulp
// A component below 1e-30 does not move the ray (t <= ~1e12): the slab test of that axis must come out as "lo <= o <= hi", planes included, for every t.  With +1e30
// WHATEVER THE SIGN it does: every quantised plane lies at least 1e-3 quantum outside its box (bvh_build.hip grid_origin, k_collapse), so an origin on a face is
// strictly between the planes and both distances are huge with opposite signs.  (-1e30 for a tiny negative component, with the lower faces still exact, turned "on the
// lower face" into "left at t = 0": a lost edge hit — tests/test_gpu_parity.py::test_lattice_rays, direction components of -1e-45.)
__device__ __forceinline__ float safe_inv(float d) { return absf(d) < 1e-30f ? 1e30f : __builtin_amdgcn_rcpf(d); }

struct Hit { uint32_t inst, tri; float t, u, v; };   // tri = slot of the hit triangle's record in SceneView::tris

// per-lane traversal state
struct Lane {
    f3 o, id;          // current-space ray origin and reciprocal direction
    RayK rk;
    Hit best;
    uint32_t g0, g1;   // current child group: g0 = base index, g1 = hit bits (0..7) | slot-type bits (8..15) | kind (16..17)
    uint32_t ta0, ta1; // triangle group under test (hit leaves of a visited BLAS node): item base, hit bits | lmask << 8
    uint32_t tb0, tb1; // a second pending triangle group: node traversal runs ahead of the triangle tests by up to two nodes
    uint32_t octbase;  // ray octant << 8 (row of the order table)
    uint32_t cur_inst;
    int sp, sb;        // stack entries live in rows [sb, sp): sb moves up when the bottom entry is handed to an idle lane (launch tails)
    uint32_t own;      // closest-hit tails: bits 0..5 = the lane that owns this ray (itself unless this lane searches a handed-over piece), bits 8.. = pieces still out
    int ret_sp;        // two-level scenes: >= 0 while the lane is inside an instance = the stack height at which the instance's subtree is exhausted; -1 at world level
    float slk;
    f3 wo, wid; uint32_t woct;   // two-level scenes: the world-space origin / reciprocal direction / octant, put aside while the lane is inside an instance
};

struct StackRef { uint32_t* lds; uint32_t* spill; uint32_t spill_stride; uint32_t* overflow; };

__device__ __forceinline__ void lane_push(Lane& L, const StackRef& S, uint32_t e0, uint32_t e1) {   // general push (any depth)
    if (L.sp < STACK_LDS) { S.lds[(2 * L.sp) * TRACE_BLOCK + threadIdx.x] = e0; S.lds[(2 * L.sp + 1) * TRACE_BLOCK + threadIdx.x] = e1; }
    else if (L.sp < STACK_LDS + STACK_SPILL) {
        S.spill[(size_t)(2 * (L.sp - STACK_LDS)) * S.spill_stride] = e0; S.spill[(size_t)(2 * (L.sp - STACK_LDS) + 1) * S.spill_stride] = e1;
    } else { *S.overflow = 1u; return; }
    L.sp++;
}
__device__ __forceinline__ void lane_pop(Lane& L, const StackRef& S) {
    L.sp--;
    // the LDS entry is read unconditionally (two ds_read) and overwritten in the rare spill case: written as if / else the compiler selects between the two
    // ADDRESSES and issues one flat_load pair for both — every pop then goes through the flat path
    const int e = L.sp < STACK_LDS ? L.sp : STACK_LDS - 1;
    L.g0 = S.lds[(2 * e) * TRACE_BLOCK + threadIdx.x]; L.g1 = S.lds[(2 * e + 1) * TRACE_BLOCK + threadIdx.x];
    if (__builtin_expect(L.sp >= STACK_LDS, 0)) {
        L.g0 = S.spill[(size_t)(2 * (L.sp - STACK_LDS)) * S.spill_stride]; L.g1 = S.spill[(size_t)(2 * (L.sp - STACK_LDS) + 1) * S.spill_stride];
    }
}

// Culling against the best hit so far needs an ABSOLUTE slack next to the box test's relative one: the watertight test computes t as a weighted mean of the three
// distances (in t) from the origin to the planes through the vertices across the ray's dominant axis, so it carries an error of a few ulps of the LARGEST of them —
// not of t.  A ray that starts 1e-5 in front of a large triangle gets t with a relative error of 1e-2, the slab distance of that triangle's flat box is exact, and of
// two coplanar triangles the second one's box would be culled against the first one's t although its own t is smaller by rounding (found by the randomized scenes of
// tests/test_gpu_parity.py, 5 of 20 000).  Every such distance is at most (largest coordinate of the scene + largest coordinate of the origin) / largest component
// of the direction.  With every operation rounded once the error is at most 10 ulps (6e-7) of that distance; 1.5e-6 of a bound that is itself twice too large for
// most rays.  (4e-6 was measured first: on S1 it reaches past the ~1.2e-4 by which a light sample's shadow ray stops short of the emitter, every such ray then visits
// the emitter's box and tests its triangles — k_trace_shadow +3 %.)  Per ray and space, kept in a register: recomputed at every node it costs 0.3 % more
// (profiles/r04_cull_slack.txt).
#ifndef TRACE_CULL_SLACK
#define TRACE_CULL_SLACK 1   // (0: measurements only)
#endif
__device__ __forceinline__ float cull_slack(const Lane& L, float coord_slack) {
    if (!TRACE_CULL_SLACK) return 0.0f;
    return __builtin_fmaf(fmaxf(fmaxf(absf(L.o.x), absf(L.o.y)), absf(L.o.z)), 1.5e-6f, coord_slack) * fminf(fminf(absf(L.id.x), absf(L.id.y)), absf(L.id.z));
}
__device__ __forceinline__ void lane_set_space(Lane& L, f3 o, f3 d, float coord_slack) {
    L.o = o;
    L.id = F3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    L.slk = cull_slack(L, coord_slack);
    L.rk = rayk_make(o, d);
    L.octbase = ((L.id.x < 0.0f ? 1u : 0u) | (L.id.y < 0.0f ? 2u : 0u) | (L.id.z < 0.0f ? 4u : 0u)) << 8;
}

__device__ __forceinline__ bool lane_begin(Lane& L, const SceneView& sc, f3 o, f3 d, float tmax) {
    lane_set_space(L, o, d, sc.coord_slack);
    L.best.inst = MAX_UINT; L.best.tri = 0; L.best.t = tmax; L.best.u = 0.0f; L.best.v = 0.0f;
    L.sp = 0; L.sb = 0; L.ret_sp = -1; L.cur_inst = WORLD_INSTANCE;   // (cur_inst: only used when the root IS the world BLAS)
    L.g0 = sc.tlas_root; L.g1 = sc.tlas_root != MAX_UINT ? (GRP_NODE | 0x0101u) : 0u;   // a group of one: the root itself
    L.ta0 = 0; L.ta1 = 0; L.tb0 = 0; L.tb1 = 0;
    return sc.tlas_root != MAX_UINT;
}

// ---- the step bodies.  The wave loop decides, per iteration, which bodies run (see trace_wave_loop). ----

// Takes the next child of the lane's group in octant order; what is left of the group goes back on the stack.
// Returns the child's index (node index for GRP_NODE, TLAS item for GRP_INST).
__device__ __forceinline__ uint32_t group_take(Lane& L, const StackRef& S, const uint8_t* lut) {
    const uint32_t hits = L.g1 & 0xffu;
    const uint32_t s = lut[L.octbase + hits];
    const uint32_t bit = 1u << s;
    const uint32_t rest = L.g1 & ~bit;
    const uint32_t idx = L.g0 + __popc((L.g1 >> 8) & 0xffu & (bit - 1u));
    if (__builtin_expect(L.sp >= STACK_LDS, 0)) {
        if (rest & 0xffu) lane_push(L, S, L.g0, rest);
    } else {   // branch-free: the write always happens, the stack pointer only moves if something is left
        S.lds[(2 * L.sp) * TRACE_BLOCK + threadIdx.x] = L.g0; S.lds[(2 * L.sp + 1) * TRACE_BLOCK + threadIdx.x] = rest;
        L.sp += (rest & 0xffu) ? 1 : 0;
    }
    L.g1 = 0u;
    return idx;
}

// internal node: 5 x 16-B loads, 8 quantised box tests → 8 hit bits.  Straight-line code, no per-child entries:
// the hit internal children become the lane's new group, the hit leaves its leaf group (BLAS) or an instance group (TLAS).
template <bool STATS, bool ANY_HIT>
__device__ __forceinline__ void step_node(Lane& L, const SceneView& sc, const StackRef& S, uint32_t node, bool in_blas /* the node belongs to a BLAS: its leaves are triangles */, unsigned long long& nv) {
    const uint4* np = reinterpret_cast<const uint4*>(sc.nodes + node);
    const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3], w4 = np[4];
    if (STATS) nv++;
    const float nox = u2f(w0.x), noy = u2f(w0.y), noz = u2f(w0.z);
    const uint32_t ex = w0.w & 0xff, ey = (w0.w >> 8) & 0xff, ez = (w0.w >> 16) & 0xff, imask = w0.w >> 24;
    const float ax = u2f(ex << 23) * L.id.x, ay = u2f(ey << 23) * L.id.y, az = u2f(ez << 23) * L.id.z;
    // (+0 through the SAME instruction: the hit bit below is a sign bit, and a ray that starts exactly in the plane of a flat box against its direction has a plane
    // distance of 0 * negative = -0, whose sign would read as a miss — the oracle's `tf >= 0` takes it: tests/test_gpu_parity.py::test_rays_at_the_hulls_..., seed 14)
    float bx = __builtin_fmaf(nox - L.o.x, L.id.x, 0.0f), by = __builtin_fmaf(noy - L.o.y, L.id.y, 0.0f), bz = __builtin_fmaf(noz - L.o.z, L.id.z, 0.0f);
    // A distance to the grid's origin plane that OVERFLOWED says nothing — and as +-infinity it would say something: with a finite step a every plane of the axis then
    // lies at -infinity (exit before entry: a miss) although the true distances are finite.  It happens where a node is astronomically large against the ray's
    // reciprocal direction: an instance whose culling pad came out as 1e37 (a sheared transform that loses every digit, round 6's sweep: tests/test_gpu_parity.py
    // ::test_rays_at_the_hulls_..., seeds 6709891 and 6711985 — 19 of 180 rays lost every instance that shared a TLAS node with it), coordinates beyond 3e8 under an
    // axis-parallel ray (1 / d = 1e30).  x * 0 + x is x for every finite x and not a number for an infinite one: the axis then decides nothing (v_min / v_max pass
    // the other operand on), which is the conservative answer.
#if !defined(TRACE_B_OVERFLOW_IS_INFINITE)   // (measurements only: the kernels without the three instructions)
    bx = __builtin_fmaf(bx, 0.0f, bx); by = __builtin_fmaf(by, 0.0f, by); bz = __builtin_fmaf(bz, 0.0f, bz);
#endif
    // byte planes: qlo[0] = w2.xy, qlo[1] = w2.zw, qlo[2] = w3.xy, qhi[0] = w3.zw, qhi[1] = w4.xy, qhi[2] = w4.zw.
    // a = scale * id has the sign of id, so the entry plane of an axis is qlo when id >= 0 and qhi otherwise
    // (the choice made by ADDRESS instead — a 128-B node on a 128-B boundary with every axis' planes as {lo, hi, lo}, three 16-B loads at offset 0 or 8 by the ray's
    // direction signs: 7 VALU fewer per visit and one cache line per node instead of 1.5 — was built and measured: S1 -9 %, S2 -4 %, the node pool's footprint
    // (+60 %) costs far more than the selects: profiles/r05_tri_density.txt)
    const bool sx = L.id.x < 0.0f, sy = L.id.y < 0.0f, sz = L.id.z < 0.0f;
    const uint32_t nx[2] = { sx ? w3.z : w2.x, sx ? w3.w : w2.y }, fx[2] = { sx ? w2.x : w3.z, sx ? w2.y : w3.w };
    const uint32_t ny[2] = { sy ? w4.x : w2.z, sy ? w4.y : w2.w }, fy[2] = { sy ? w2.z : w4.x, sy ? w2.w : w4.y };
    const uint32_t nz[2] = { sz ? w4.z : w3.x, sz ? w4.w : w3.y }, fz[2] = { sz ? w3.x : w4.z, sz ? w3.y : w4.w };
    // (v_pk_fma_f32 for the {entry, exit} pairs was measured: 24 instructions fewer per node, 3 % slower overall)
    // (cull_slack: the absolute part of the slack; the relative part is the 1.00001 below.  An any-hit ray's tmax never changes: the sum is made once per
    // space and kept where a closest-hit ray keeps u — any_hit_cull)
    const float tlimit = ANY_HIT ? L.best.u : L.best.t + L.slk;
    // hit <=> f * 1.00001 - n >= 0.  On gfx950 only v_fma / v_mul / v_add / v_sub (f32) and a few integer ops issue in 2 cycles per wave, everything else —
    // compares, selects, min / max, conversions — in 4 (profiles/r03_valu_microbench.txt): ONE fma, whose sign bit ONE v_alignbit shifts into a mask of
    // MISS bits, replaces mul + cmp + cndmask + or.  Children 7..0, so that child 0 ends up in bit 0.  (A NaN can only come from a NaN ray; either sign
    // is fine for it: boxes only gate which triangles are tested.)
    uint32_t miss8 = 0;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        const int wi = i >> 2, sh = (i & 3) * 8;
        const float t0x = __builtin_fmaf((float)((nx[wi] >> sh) & 0xff), ax, bx), t1x = __builtin_fmaf((float)((fx[wi] >> sh) & 0xff), ax, bx);
        const float t0y = __builtin_fmaf((float)((ny[wi] >> sh) & 0xff), ay, by), t1y = __builtin_fmaf((float)((fy[wi] >> sh) & 0xff), ay, by);
        const float t0z = __builtin_fmaf((float)((nz[wi] >> sh) & 0xff), az, bz), t1z = __builtin_fmaf((float)((fz[wi] >> sh) & 0xff), az, bz);
        const float n = fmaxf(fmaxf(t0x, t0y), fmaxf(t0z, 0.0f));
        const float f = fminf(fminf(t1x, t1y), fminf(t1z, tlimit));
        miss8 = __builtin_amdgcn_alignbit(miss8, f2u(__builtin_fmaf(f, 1.00001f, -n)), 31);   // (miss8 << 1) | sign
    }
    const uint32_t hits8 = ~miss8 & 0xffu;
#if defined(TRACE_COUNT_EMPTY)   // measurement: node visits none of whose children the ray hits — all of them (1), those made with a hit already found (2), or (3) those
    // that a bound carried in the stack entry could have culled at the pop: the NODE'S OWN entry distance (its grid from the origin plane to plane 255, the side the
    // ray enters by) lies beyond the limit the visit tested against — in the upper half of the visit counter (tools/empty_visits.py, profiles/r06_stale_visits.txt)
    if (TRACE_COUNT_EMPTY == 3) {
        const float ex_ = __builtin_fmaf(L.id.x < 0.0f ? 255.0f : 0.0f, ax, bx), ey_ = __builtin_fmaf(L.id.y < 0.0f ? 255.0f : 0.0f, ay, by), ez_ = __builtin_fmaf(L.id.z < 0.0f ? 255.0f : 0.0f, az, bz);
        if (STATS && fmaxf(fmaxf(ex_, ey_), ez_) > tlimit * 1.00001f) nv += (1ull << 32);
    } else
    if (STATS && !(hits8 & (w0.w >> 24 | (w1.z & 0xffu))) && (TRACE_COUNT_EMPTY == 1 || (!ANY_HIT && L.best.inst != MAX_UINT))) nv += (1ull << 32);
#endif
    // empty slots (inverted boxes) can pass the slack test when the node is tiny against its distance: imask / lmask drop them
    const uint32_t lmask = w1.z & 0xffu;
    const uint32_t ihits = hits8 & imask, lhits = hits8 & lmask;
    L.g0 = w1.x; L.g1 = GRP_NODE | (imask << 8) | ihits;
    if (in_blas) {   // hit leaves = one triangle each: they queue up behind the group under test (the caller guarantees tb is free)
        const bool a_free = (L.ta1 & 0xffu) == 0u;
        const uint32_t n0 = w1.y, n1 = (lmask << 8) | lhits;
        L.tb0 = a_free ? L.tb0 : n0; L.tb1 = a_free ? L.tb1 : n1;
        L.ta0 = a_free ? n0 : L.ta0; L.ta1 = a_free ? n1 : L.ta1;
    } else if (lhits) {   // TLAS: hit leaves are instances; visit them before the internal children
        if (ihits) lane_push(L, S, L.g0, L.g1);
        L.g0 = w1.y; L.g1 = GRP_INST | (lmask << 8) | lhits;
    }
}

// ONE triangle of the group under test per call (any order: every hit leaf of a visited node is tested).
// Returns true when an any-hit ray is finished.
template <bool ANY_HIT, bool STATS, bool INSTANCED>
__device__ __forceinline__ bool step_tri(Lane& L, const SceneView& sc, unsigned long long& nt) {
    const uint32_t low = L.ta1 & (0u - L.ta1);   // lowest hit bit (the caller guarantees there is one)
    const uint32_t idx = L.ta0 + __popc((L.ta1 >> 8) & (low - 1u));
    L.ta1 &= ~low;
    if (!(L.ta1 & 0xffu)) { L.ta0 = L.tb0; L.ta1 = L.tb1; L.tb1 = 0u; }
    // three 12-B loads whose ADDRESS carries the rotation by the ray's dominant axis (msne_device.h TriRot), and the instance of a world-BLAS triangle from the same 64-B record
    const float* tp = reinterpret_cast<const float*>(sc.tri_rot + idx) + L.rk.kz;
    typedef float f3v __attribute__((ext_vector_type(3)));
    f3v a, b, c;
    __builtin_memcpy(&a, tp, 12); __builtin_memcpy(&b, tp + 5, 12); __builtin_memcpy(&c, tp + 10, 12);
    uint32_t winst = WORLD_INSTANCE;   // (an any-hit ray of a scene without a TLAS level only needs "hit")
    if (INSTANCED ? L.cur_inst == WORLD_INSTANCE : !ANY_HIT) winst = sc.tri_rot[idx].inst;   // (inside an instance of its own the lane knows the instance: no load)
    if (STATS) nt++;
    float t, u, v;
    RayK rk = L.rk;
    if (INSTANCED && !TRACE_KEEP_RO_TLAS) rk.ro = rot3(L.o, L.rk.kz);   // (two-level scenes: three registers fewer across the loop for six selects per test)
    const bool hit = tri_intersect(rk, F3(a.x, a.y, a.z), F3(b.x, b.y, b.z), F3(c.x, c.y, c.z), t, u, v);
#if defined(TRACE_TRI_DUP)   // measurement only (profiles/r05_tri_density.txt): the triangle test issued twice — what one more pass of the body costs
    { RayK k2 = rk; asm volatile("" : "+v"(k2.ro.x), "+v"(k2.ro.y), "+v"(k2.ro.z));
      float t2, u2, v2;
      const bool h2 = tri_intersect(k2, F3(a.x, a.y, a.z), F3(b.x, b.y, b.z), F3(c.x, c.y, c.z), t2, u2, v2);
      if (h2 != hit) { t = t2; u = u2; v = v2; } }   // (never: the same inputs)
#endif
    const uint32_t inst = (!INSTANCED || L.cur_inst == WORLD_INSTANCE) ? winst : L.cur_inst;   // world BLAS (the only one of a scene without a TLAS level): the triangle record names its instance
    if (ANY_HIT) {
        const bool done = hit && t < L.best.t;
        L.best.inst = done ? inst : L.best.inst;
        return done;
    }
    bool closer = hit && t < L.best.t;
    if (__builtin_expect(hit && t == L.best.t && L.best.inst != MAX_UINT, 0)) {   // exact tie: smallest (instance, geometry, primitive) wins
        const TriRec* bt = sc.tris + L.best.tri; const TriRec* ct = sc.tris + idx;
        const uint32_t bgeo = bt->geo, bprim = bt->prim, cgeo = ct->geo, cprim = ct->prim;
        closer = inst < L.best.inst || (inst == L.best.inst && (cgeo < bgeo || (cgeo == bgeo && cprim < bprim)));
    }
    L.best.t = closer ? t : L.best.t; L.best.u = closer ? u : L.best.u; L.best.v = closer ? v : L.best.v;
    L.best.inst = closer ? inst : L.best.inst; L.best.tri = closer ? idx : L.best.tri;
    return false;
}

// ---------------------------------------------------------------------------------------------
// Persistent-wave dequeue.  One device-scope atomic word saturates at ~88 dequeues/us on MI355X
// (MI355X_MICROARCH.md "dequeue"), so (1) every wave's FIRST chunk is static (wave w takes chunk w: an
// empty or short queue costs no atomics at all), (2) later chunks come from one atomicAdd per wave on the
// queue head, and (3) a chunk is 64..256 rays that the wave's lanes consume one by one as they go idle.
// Small chunks handed out IN ORDER also keep all resident waves inside one window of ~1.5 M consecutive rays of the
// (screen-ordered) queue, so they share BVH nodes in L2: a big static share per wave was measured 1.4-2x slower.
// (An XCD-aware hand-out — the queue cut into eight contiguous ranges, the waves of XCD b % 8 draining range x from a head of its own, a dry range's waves moving to the
// fullest one — was built and measured in round 5: S1 -2 %, 20-launch batches -3.5 %, S2 -5 %.  One window of the whole queue shared by all eight L2s beats eight windows
// with an eighth of the screen each: profiles/r05_tri_density.txt section 6.)
struct WaveQueue {
    uint32_t n, chunk, pos, end, nwaves_chunk;
    uint32_t* head;
    bool exhausted;
    __device__ WaveQueue(uint32_t n_, uint32_t* head_) : n(n_), head(head_), exhausted(false) {
        const uint32_t nwaves = gridDim.x * (TRACE_BLOCK / 64);
        uint32_t c = (n / (nwaves * 4u) + 32u) & ~63u;   // (16-ray chunks for short queues were measured: more, emptier waves — slower)
        // What the static first chunks do not cover is drained through ONE atomic word: with 64-ray chunks that caps a launch at ~5.6 Grays/s, so a queue
        // that needs the atomic at all is cut into chunks of at least 128 (0.8 M any-hit rays: 223 -> 160 us, 1.6 M: 316 -> 234; an 8-way shard of 20 launches 6.01 -> 5.76 ms)
        const uint32_t lo = n <= nwaves * 64u ? 64u : 128u;
        chunk = c < lo ? lo : (c > 256u ? 256u : c);   // 256: a launch ends when its last wave ends, one chunk (~0.2 ms) after the first
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

// lane index of the r-th set bit of a 64-lane mask (r < popcount)
__device__ __forceinline__ uint32_t nth_set_bit(unsigned long long mask, uint32_t r) {
    uint32_t m = (uint32_t)mask, base = 0, c = (uint32_t)__popc(m);
    if (r >= c) { r -= c; m = (uint32_t)(mask >> 32); base = 32; }
    c = (uint32_t)__popc(m & 0xffffu); if (r >= c) { r -= c; m >>= 16; base += 16; } m &= 0xffffu;
    c = (uint32_t)__popc(m & 0xffu);   if (r >= c) { r -= c; m >>= 8;  base += 8; }  m &= 0xffu;
    c = (uint32_t)__popc(m & 0xfu);    if (r >= c) { r -= c; m >>= 4;  base += 4; }  m &= 0xfu;
    c = (uint32_t)__popc(m & 0x3u);    if (r >= c) { r -= c; m >>= 2;  base += 2; }  m &= 0x3u;
    return base + ((r >= (m & 1u)) ? 1u : 0u);
}

// visit-order table: lut[octant << 8 | hit bits] = the hit slot s with the smallest (s ^ octant) — slot == octant is the
// child the ray enters first, slot == ~octant the one it reaches last (the builder places children in octant order)
__device__ __forceinline__ void order_table_init(uint8_t* lut) {
    for (uint32_t e = threadIdx.x; e < 2048u; e += TRACE_BLOCK) {
        const uint32_t oct = e >> 8, mask = e & 255u;
        uint32_t best = 0, bestkey = 8;
        for (uint32_t s = 0; s < 8; s++) if (((mask >> s) & 1u) && (s ^ oct) < bestkey) { bestkey = s ^ oct; best = s; }
        lut[e] = (uint8_t)best;
    }
    __syncthreads();
}

// Wave loop shared by the three kernels.  `load(i, o, d, tmax)` returns false for entries without a ray;
// `store(i, lane)` receives the finished lane.
// INSTANCED = the scene has a TLAS level.  Without one (sc.root_in_blas: the world BLAS is the root) no lane ever changes space,
// and that instantiation carries none of the space-body bookkeeping (measured on S1: the shared code cost 7 %).
template <bool ANY_HIT, bool STATS, bool INSTANCED, class Load, class LoadDir, class Store>
__device__ __forceinline__ void trace_wave_loop(const SceneView& sc, uint32_t n, uint32_t* head, uint32_t* lds_stack, const uint8_t* lut, uint32_t* spill, uint32_t* overflow,
                                                uint32_t refill /* idle lanes of 64 that trigger a refill from the ray queue */, Load load, LoadDir load_dir /* (i) -> the ray's direction again */, Store store, unsigned long long& nv, unsigned long long& nt, unsigned long long* prof,
                                                uint32_t* rays_traced = nullptr /* += queue entries that held a ray */, unsigned long long* lanes_out = nullptr /* STATS: lane use, 12 counters */) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t gtid = blockIdx.x * TRACE_BLOCK + threadIdx.x;
    StackRef S{ lds_stack, spill + gtid, gridDim.x * TRACE_BLOCK, overflow };
    WaveQueue wq(n, head);
    Lane L; L.sp = 0; L.sb = 0; L.g1 = 0; L.ta1 = 0; L.tb1 = 0; L.own = lane;
    bool active = false; uint32_t my = 0, n_rays = 0;
    // STATS builds: wave-cycle profile of the loop sections (s_memtime), accumulated per wave
    unsigned long long cyc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tprev = 0;
    // lane use per iteration (MsneGetTraversalLaneUse): [0] iterations that reach the bodies, [1] lanes with a ray, [2] / [3] / [4] lanes in the node / triangle / space
    // body, [5] lanes whose group is an instance while the space body does not run, [6] lanes with a ray that run no body at all, [7] / [8] / [9] iterations that
    // run the node / triangle / space body, [10] lanes that hold a node group but wait for their triangle queue to drain
    unsigned long long use[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    auto lap = [&](int k) { if (STATS) { const unsigned long long t = __builtin_readcyclecounter(); cyc[k] += t - tprev; tprev = t; } };
    if (STATS) tprev = __builtin_readcyclecounter();
    for (;;) {
        if (STATS) cyc[7] += 1;   // iterations
        // (a) lanes without a child group pop one; a lane with nothing left at all is finished
        bool piece_done = false;   // closest-hit tails: this lane finished a handed-over piece of another lane's ray
        if (active && !(L.g1 & 0xffu)) {
            const bool has_t = (L.ta1 & 0xffu) != 0u;
            // leaving an instance needs no stack entry and no body: when the stack is back at the height of the entry the instance's subtree is exhausted, and —
            // once the triangles queued in instance space are tested — the lane takes back the world-space half of its ray (the shear constants stay stale:
            // every entry recomputes them) and pops at TLAS level in the same iteration
            bool wait_t = false;
            if (INSTANCED && L.ret_sp >= 0 && L.sp == L.ret_sp) {
                if (has_t) wait_t = true;
                else { L.o = L.wo; L.id = L.wid; L.octbase = L.woct; L.ret_sp = -1; L.slk = cull_slack(L, sc.coord_slack); if (ANY_HIT) L.best.u = L.best.t + L.slk; }
            }
            if (!wait_t) {
                if (L.sp == L.sb) {
                    if (!has_t) {
                        if (ANY_HIT || L.own == lane) { store(my, L); active = false; }   // (an owner with pieces out — own >> 8 != 0 — waits for them)
                        else if ((L.own & 63u) != lane) piece_done = true;
                    }
                } else lane_pop(L, S);
            }
        }
        if (!ANY_HIT) {   // finished pieces report to the lane that owns the ray: its best hit absorbs theirs (all in one wave: no atomics)
            unsigned long long fm = __ballot(piece_done);
            while (fm) {
                const int f = __builtin_ctzll(fm); fm &= fm - 1ull;
                const uint32_t o_ = (uint32_t)__builtin_amdgcn_readlane((int)L.own, f) & 63u;
                const uint32_t c_inst = (uint32_t)__builtin_amdgcn_readlane((int)L.best.inst, f), c_tri = (uint32_t)__builtin_amdgcn_readlane((int)L.best.tri, f);
                const float c_t = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(L.best.t), f)), c_u = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(L.best.u), f)), c_v = u2f((uint32_t)__builtin_amdgcn_readlane((int)f2u(L.best.v), f));
                if (lane == o_) {
                    L.own -= 1u << 8;
                    bool closer = c_inst != MAX_UINT && c_t < L.best.t;
                    if (c_inst != MAX_UINT && c_t == L.best.t && L.best.inst != MAX_UINT) {   // exact tie: smallest (instance, geometry, primitive) wins
                        const TriRec* a = sc.tris + c_tri; const TriRec* b = sc.tris + L.best.tri;
                        closer = c_inst < L.best.inst || (c_inst == L.best.inst && (a->geo < b->geo || (a->geo == b->geo && a->prim < b->prim)));
                    }
                    L.best.t = closer ? c_t : L.best.t; L.best.u = closer ? c_u : L.best.u; L.best.v = closer ? c_v : L.best.v;
                    L.best.inst = closer ? c_inst : L.best.inst; L.best.tri = closer ? c_tri : L.best.tri;
                }
            }
            if (piece_done) active = false;
        }
        lap(0);
        // (b) idle lanes take new rays from the wave's private chunk
        const unsigned long long act = __ballot(active);
        const uint32_t nidle = 64u - (uint32_t)__popcll(act);   // blocks are 4 full waves
        if (!wq.exhausted && nidle >= refill) {
            uint32_t base;
            const uint32_t got = wq.take(nidle, base);
            const uint32_t r = (uint32_t)__popcll(~act & lt);
            if (!active && r < got) {
                my = base + r;
                f3 o, d; float tmax;
                if (load(my, o, d, tmax)) { n_rays++; L.own = lane; active = lane_begin(L, sc, o, d, tmax); if (ANY_HIT) L.best.u = tmax + L.slk; if (!active) store(my, L); }
                else { L.best.inst = MAX_UINT; L.best.tri = 0; L.best.u = 0.0f; L.best.v = 0.0f; L.best.t = 0.0f; store(my, L); }
            }
        }
        // (b') launch tails: a launch ends with a few very long rays (S1: mean 16-20 iterations, max 185 closest / 330 shadow)
        // run by a few lanes at memory latency.  Once the queue is empty, a busy lane hands the BOTTOM entry of its stack —
        // the largest unvisited subtree — to an idle lane of its wave, which searches it as a ray of its own for the same
        // queue entry.  Any-hit: occlusion is an OR, the pieces never meet again.  Closest-hit: a finished piece reports
        // to the lane that owns the ray (see (a)); the owner stores the hit when no piece is out.
        if (wq.exhausted) {
            const unsigned long long busy = __ballot(active);
            const bool give = active && L.sp > L.sb && L.ret_sp < 0;   // only entries that live in world space
            const unsigned long long donors = __ballot(give), idle = ~busy;
            const uint32_t nthief = (uint32_t)__popcll(idle), ndon = (uint32_t)__popcll(donors);
            if (ndon != 0u && nthief >= 8u) {
                const uint32_t npair = nthief < ndon ? nthief : ndon;
                const uint32_t rth = (uint32_t)__popcll(idle & lt), rdn = (uint32_t)__popcll(donors & lt);
                const uint32_t dl = nth_set_bit(donors, rth < npair ? rth : 0u);   // the donor this lane would take from
                const float ox = __shfl(L.o.x, dl), oy = __shfl(L.o.y, dl), oz = __shfl(L.o.z, dl);
                const float ix = __shfl(L.id.x, dl), iy = __shfl(L.id.y, dl), iz = __shfl(L.id.z, dl);
                const uint32_t kz = (uint32_t)__shfl((int)L.rk.kz, dl);
                const float sx = __shfl(L.rk.S1, dl), sy = __shfl(L.rk.S2, dl), sz = __shfl(L.rk.Sz, dl);
                // the donor's best hit so far goes along whole: a piece that meets the same distance again (coincident triangles) must break the tie against it
                const float bt = __shfl(L.best.t, dl), bu = __shfl(L.best.u, dl), bv = __shfl(L.best.v, dl);
                const uint32_t binst = (uint32_t)__shfl((int)L.best.inst, dl), btri = (uint32_t)__shfl((int)L.best.tri, dl);
                const uint32_t ob = __shfl(L.octbase, dl), ci = __shfl(L.cur_inst, dl), dmy = __shfl(my, dl);
                const int dsb = __shfl(L.sb, dl);
                const uint32_t downer = (uint32_t)__shfl((int)L.own, dl) & 63u;   // the donor may itself be searching a piece
                const bool thief = !active && rth < npair;
                if (thief) {
                    const uint32_t col = threadIdx.x - lane + dl;   // the donor's stack column
                    if (dsb < STACK_LDS) { L.g0 = S.lds[(2 * dsb) * TRACE_BLOCK + col]; L.g1 = S.lds[(2 * dsb + 1) * TRACE_BLOCK + col]; }
                    else {
                        const uint32_t* dsp = S.spill - lane + dl;
                        L.g0 = dsp[(size_t)(2 * (dsb - STACK_LDS)) * S.spill_stride]; L.g1 = dsp[(size_t)(2 * (dsb - STACK_LDS) + 1) * S.spill_stride];
                    }
                    L.o = F3(ox, oy, oz); L.id = F3(ix, iy, iz); L.slk = cull_slack(L, sc.coord_slack);
                    L.rk.kz = kz; L.rk.S1 = sx; L.rk.S2 = sy; L.rk.Sz = sz; L.rk.ro = rot3(L.o, kz);
                    L.best.inst = binst; L.best.tri = btri; L.best.t = bt; L.best.u = ANY_HIT ? bt + L.slk : bu; L.best.v = bv;
                    L.octbase = ob; L.cur_inst = ci; L.ret_sp = -1; my = dmy;
                    L.sp = 0; L.sb = 0; L.ta1 = 0; L.tb1 = 0; L.own = downer;
                    active = true;
                }
                if (give && rdn < npair) L.sb++;
                if (!ANY_HIT) {   // one more piece out for every owner that was taken from
                    unsigned long long tm = __ballot(thief);
                    while (tm) {
                        const int tl = __builtin_ctzll(tm); tm &= tm - 1ull;
                        const uint32_t o_ = (uint32_t)__builtin_amdgcn_readlane((int)L.own, tl) & 63u;
                        if (lane == o_) L.own += 1u << 8;
                    }
                }
            }
        }
        lap(1);
        if (!__ballot(active)) { if (wq.exhausted) break; continue; }
        // (c) both bodies run whenever any lane has work for them (vote thresholds were swept: within noise).  Node traversal
        // may run ahead of the triangle tests by two nodes' worth of hit leaves; visiting nodes with a not yet shortened
        // ray is conservative (a few extra visits), results do not change.
        const bool want_t = active && (L.ta1 & 0xffu);
        const bool has_g = active && (L.g1 & 0xffu);
        const bool want_n = has_g && (!INSTANCED || (L.g1 & GRP_KIND_MASK) == GRP_NODE) && !(L.tb1 & 0xffu);
        const bool want_s = INSTANCED && has_g && (L.g1 & GRP_KIND_MASK) == GRP_INST;   // enters an instance: the ray changes space
        const unsigned long long mt = __ballot(want_t);
        const bool do_n = __ballot(want_n) != 0ull;
        // (TRACE_TRI_MIN_LANES > 1, measurement only: the triangle body waits until that many lanes want it or no lane can visit a node — profiles/r05_tri_density.txt)
        const bool do_t = TRACE_TRI_MIN_LANES <= 1 ? mt != 0ull : (mt != 0ull && ((uint32_t)__popcll(mt) >= (uint32_t)TRACE_TRI_MIN_LANES || !do_n));
        // Changing space costs ~250 instructions (reload the world-space ray, transform, three IEEE divisions for the shear constants)
        // and few lanes need it in any one iteration: entering and leaving share one body, and it runs when enough lanes wait for
        // it or nothing else can be done.
        const unsigned long long ms = INSTANCED ? __ballot(want_s) : 0ull;
        const bool do_s = INSTANCED && ms != 0ull && ((uint32_t)__popcll(ms) >= SPACE_MIN_LANES || (!do_n && !do_t));
        if (STATS && do_s) cyc[2] += __popcll(ms);   // lanes that change space (slot 2 of the profile: the vote itself is timed with the node section)
        if (INSTANCED && do_s && want_s) {   // entering an instance (leaving needs no body: the world-space half of the lane's ray was put aside)
            const uint32_t item = group_take(L, S, lut);   // TLAS leaf (one instance of the group; the rest of the group goes back on the stack)
            // the leaf's own 80-B record (matrix, BLAS root, instance, bounding sphere) and the ray's direction: six independent loads, one round trip
            const float4* mp = reinterpret_cast<const float4*>(sc.tlas_leaves + item);
            const float4 r0 = mp[0], r1 = mp[1], r2 = mp[2], sph = mp[4];
            const uint4 lw = reinterpret_cast<const uint4*>(mp)[3];
            const uint32_t new_inst = lw.y, flags = lw.z;
            uint32_t root = lw.x;
            f3 o = L.o, d = load_dir(my);   // at TLAS level the lane's origin IS the world-space one; the world-space direction is not kept in registers
            {   // The instance's world-space bounding sphere (msne_device.h TlasLeaf), before the change of space and the visit of the BLAS root are paid for: a ray whose
                // line misses the sphere, or that starts outside it and points away, cannot meet any of the instance's triangles.  Only ever a cull: the sphere is taken
                // 0.2 % larger than the farthest vertex, plus four times the absolute slack the tmax cull uses (the rounding of centre - origin is an absolute error of the
                // size of the coordinates, not of the radius), and the discriminant gets a tolerance ten times its rounding error: a near miss is an entry.
                const f3 oc = F3(sph.x - o.x, sph.y - o.y, sph.z - o.z);
                const float rr = __builtin_fmaf(sph.w, 1.002f, 4.0f * __builtin_fmaf(fmaxf(fmaxf(absf(o.x), absf(o.y)), absf(o.z)), 1.5e-6f, sc.coord_slack));
                const float oc2 = dot(oc, oc), dd = dot(d, d), b = dot(oc, d), r2s = rr * rr;
                const float disc = b * b - dd * (oc2 - r2s);
#if !defined(TRACE_NO_SPHERE_CULL)
                if (oc2 > r2s && (b * b + dd * oc2) < 3.0e38f && (b <= 0.0f || disc < -1e-5f * (b * b + dd * oc2))) root = MAX_UINT;   // (a sum that overflowed or is not a number decides nothing)
#endif
            }
            if (root != MAX_UINT) {
                const bool ident = (flags & INST_FLAG_IDENTITY) != 0u;
                L.wo = L.o; L.wid = L.id; L.woct = L.octbase;
                L.ret_sp = L.sp;   // (the rest of the instance group is already on the stack: group_take above)
                if (!ident) {   // t is preserved: d is not renormalised.  (Identity: M·(o,1) = o and M·d = d exactly — only the shear constants are recomputed)
                    m34 M;
                    M.m[0][0] = r0.x; M.m[0][1] = r0.y; M.m[0][2] = r0.z; M.m[0][3] = r0.w;
                    M.m[1][0] = r1.x; M.m[1][1] = r1.y; M.m[1][2] = r1.z; M.m[1][3] = r1.w;
                    M.m[2][0] = r2.x; M.m[2][1] = r2.y; M.m[2][2] = r2.z; M.m[2][3] = r2.w;
                    const f3 oi = m34_mul_point(M, o), di = m34_mul_vec(M, d);
                    o = oi; d = di;
                }
                lane_set_space(L, o, d, sc.coord_slack);
                if (ANY_HIT) L.best.u = L.best.t + L.slk;
                L.cur_inst = new_inst; L.g0 = root; L.g1 = GRP_NODE | 0x0101u;   // a group of one: the BLAS root
            }
        }
        // lanes that just entered hold their BLAS root, lanes that just left hold a TLAS group: they visit it in this iteration
        const bool want_n2 = INSTANCED && do_s ? (active && (L.g1 & 0xffu) && (L.g1 & GRP_KIND_MASK) == GRP_NODE && !(L.tb1 & 0xffu)) : want_n;
        const bool do_n2 = INSTANCED && do_s ? __ballot(want_n2) != 0ull : do_n;
        if (do_n2 && want_n2) {
            const uint32_t idx = group_take(L, S, lut);
            step_node<STATS, ANY_HIT>(L, sc, S, idx, INSTANCED ? (L.ret_sp >= 0 || sc.root_in_blas != 0u) : true, nv);
        }
        lap(3);
        if (STATS && do_n2) cyc[6] += __popcll(__ballot(want_n2));   // node-lane steps
        if (STATS) {
            const bool ran_s = INSTANCED && do_s && want_s, ran_n = do_n2 && want_n2, ran_t = do_t && want_t;
            use[0] += 1; use[1] += __popcll(__ballot(active));
            use[2] += __popcll(__ballot(ran_n)); use[3] += __popcll(__ballot(ran_t)); use[4] += __popcll(__ballot(ran_s));
            use[5] += __popcll(__ballot(want_s && !do_s)); use[6] += __popcll(__ballot(active && !ran_s && !ran_n && !ran_t));
            use[7] += do_n2 ? 1 : 0; use[8] += do_t ? 1 : 0; use[9] += (INSTANCED && do_s) ? 1 : 0;
            use[10] += __popcll(__ballot(has_g && !want_s && !want_n));
        }
        if (STATS && !INSTANCED && do_t) cyc[2] += 1;   // (scenes without a TLAS level: slot 2 counts the iterations that ran the triangle body)
        if (do_t && want_t) {
            if (step_tri<ANY_HIT, STATS, INSTANCED>(L, sc, nt)) { L.sp = 0; L.sb = 0; L.g1 = 0; L.ta1 = 0; L.tb1 = 0; store(my, L); active = false; }
        }
        lap(4);
        if (STATS) cyc[5] += __popcll(__ballot(active));   // active lanes at the end of the iteration
    }
    if (STATS && (threadIdx.x & 63u) == 0) { for (int k = 0; k < 8; k++) atomicAdd(&prof[k], cyc[k]); if (lanes_out) for (int k = 0; k < 12; k++) atomicAdd(&lanes_out[k], use[k]); }
    if (rays_traced) {   // one atomic per wave
        for (int o = 32; o >= 1; o >>= 1) n_rays += __shfl_xor(n_rays, o);
        if ((threadIdx.x & 63u) == 0 && n_rays) atomicAdd(rays_traced, n_rays);
    }
}

#define TRACE_LDS_DECL \
    __shared__ uint32_t lds_stack[2 * STACK_LDS * TRACE_BLOCK]; \
    __shared__ uint8_t lds_lut[2048]; \
    order_table_init(lds_lut)

template <bool STATS, bool INSTANCED>
__global__ __launch_bounds__(TRACE_BLOCK, INSTANCED ? TRACE_WPS_TLAS : TRACE_WPS) void k_trace_closest(SceneView sc, PathState st, HitBuf hits, BounceCounters* cnt,
                                                                uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out, uint32_t refill) {
    TRACE_LDS_DECL;
    const QueueDims qd = queue_dims<false>(cnt);   // the bounce's path queue: eight interleaved sub-queues, holes in the last tiles of the shorter ones (msne_device.h)
    const uint32_t n = qd.extent;
    unsigned long long nv = 0, nt = 0;
    trace_wave_loop<false, STATS, INSTANCED>(sc, n, &cnt->head_closest, lds_stack, lds_lut, spill, overflow, refill,
        [&](uint32_t i, f3& o, f3& d, float& tmax) -> bool {
            if (!queue_live<false>(qd, i)) return false;
            const float4 ro = nt_load(&st.ro[i]);
            if (f2u(ro.w) & PATH_FLAG_ZOMBIE) return false;
            const float4 rd = INSTANCED ? st.rd[i] : nt_load(&st.rd[i]);   // (two-level scenes read the direction again at every instance they enter: it stays cached)
            o = F3(ro.x, ro.y, ro.z); d = F3(rd.x, rd.y, rd.z); tmax = INFINITY_F;
            return true;
        },
        [&](uint32_t i) -> f3 { const float4 rd = st.rd[i]; return F3(rd.x, rd.y, rd.z); },
        [&](uint32_t i, const Lane& L) {
            nt_store(&hits.rec[i], make_uint4(L.best.inst, L.best.tri, f2u(L.best.u), f2u(L.best.v)));
        }, nv, nt, stat_out + 4, nullptr, stat_out + 20);
    if (STATS) { atomicAdd(&stat_out[0], nv); atomicAdd(&stat_out[1], nt); }
}

template <bool STATS, bool INSTANCED>
__global__ __launch_bounds__(TRACE_BLOCK, INSTANCED ? TRACE_WPS_TLAS : TRACE_WPS) void k_trace_shadow(SceneView sc, ShadowQueue q, BounceCounters* cnt,
                                                               uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out, uint32_t refill) {
    TRACE_LDS_DECL;
    const QueueDims qs = queue_dims<true>(cnt);   // the shadow queue the previous k_shade filled
    const uint32_t n = qs.extent;
    unsigned long long nv = 0, nt = 0;
    trace_wave_loop<true, STATS, INSTANCED>(sc, n, &cnt->head_shadow, lds_stack, lds_lut, spill, overflow, refill,
        [&](uint32_t i, f3& o, f3& d, float& tmax) -> bool {
            if (!queue_live<true>(qs, i)) return false;
            const float4 qo = nt_load(&q.o[i]), qdir = INSTANCED ? q.d[i] : nt_load(&q.d[i]);
            if (qo.w < 0.0f) return false;   // unused entry
            o = F3(qo.x, qo.y, qo.z); d = F3(qdir.x, qdir.y, qdir.z); tmax = qo.w;
            return true;
        },
        [&](uint32_t i) -> f3 { const float4 qd = q.d[i]; return F3(qd.x, qd.y, qd.z); },
        [&](uint32_t i, const Lane& L) {
            // ShadowIntersection::hit → lightSample.pdf = 0 (light.hlsl:75-77,154-156): the sample's contribution vanishes
            if (L.best.inst != MAX_UINT) q.c[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }, nv, nt, stat_out + 12, &cnt->n_shadow_traced, stat_out + 32);
    if (STATS) { atomicAdd(&stat_out[2], nv); atomicAdd(&stat_out[3], nt); }
}

// probe kernel for tests: arbitrary rays → hit records (closest) or occlusion flags (any)
template <bool ANY_HIT>
__global__ __launch_bounds__(TRACE_BLOCK, TRACE_WPS_TLAS) void k_trace_probe(SceneView sc, const float* rays /*7 per ray: o,d,tmax*/, uint32_t n, uint32_t* head,
                                                              uint32_t* out_ids /*4 per ray: hit,inst,geo,prim*/, float* out_tuv /*3 per ray*/,
                                                              uint32_t* spill, uint32_t* overflow, uint32_t refill) {
    TRACE_LDS_DECL;
    unsigned long long nv = 0, nt = 0;
    trace_wave_loop<ANY_HIT, false, true>(sc, n, head, lds_stack, lds_lut, spill, overflow, refill,
        [&](uint32_t i, f3& o, f3& d, float& tmax) -> bool {
            const float* r = rays + 7 * (size_t)i;
            o = F3(r[0], r[1], r[2]); d = F3(r[3], r[4], r[5]); tmax = r[6];
            return true;
        },
        [&](uint32_t i) -> f3 { const float* r = rays + 7 * (size_t)i; return F3(r[3], r[4], r[5]); },
        [&](uint32_t i, const Lane& L) {
            const bool hit = L.best.inst != MAX_UINT;
            if (ANY_HIT) {   // the pieces of a shared any-hit ray report separately: only occlusion is written (the buffers start zeroed)
                if (hit) { out_ids[4 * i] = 1u; out_ids[4 * i + 1] = L.best.inst; }
                return;
            }
            uint32_t geo = 0, prim = 0;
            if (hit) { geo = sc.tris[L.best.tri].geo; prim = sc.tris[L.best.tri].prim; }
            out_ids[4 * i] = hit ? 1u : 0u; out_ids[4 * i + 1] = L.best.inst; out_ids[4 * i + 2] = geo; out_ids[4 * i + 3] = prim;
            out_tuv[3 * i] = L.best.t; out_tuv[3 * i + 1] = L.best.u; out_tuv[3 * i + 2] = L.best.v;
        }, nv, nt, nullptr);
}

// ---------------- host launch wrappers ----------------
void launch_trace_closest(hipStream_t s, int grid, bool stats, const SceneView& sc, const PathState& st, const HitBuf& hits, BounceCounters* cnt,
                          uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out, uint32_t refill) {
    const bool inst = sc.root_in_blas == 0u;   // a TLAS level exists
    if (inst) grid = grid / TRACE_WPS * TRACE_WPS_TLAS;
    if (stats) { if (inst) hipLaunchKernelGGL((k_trace_closest<true, true>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, st, hits, cnt, spill, overflow, stat_out, refill);
                 else hipLaunchKernelGGL((k_trace_closest<true, false>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, st, hits, cnt, spill, overflow, stat_out, refill); }
    else { if (inst) hipLaunchKernelGGL((k_trace_closest<false, true>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, st, hits, cnt, spill, overflow, stat_out, refill);
           else hipLaunchKernelGGL((k_trace_closest<false, false>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, st, hits, cnt, spill, overflow, stat_out, refill); }
}
void launch_trace_shadow(hipStream_t s, int grid, bool stats, const SceneView& sc, const ShadowQueue& q, BounceCounters* cnt,
                         uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out, uint32_t refill) {
    const bool inst = sc.root_in_blas == 0u;
    if (inst) grid = grid / TRACE_WPS * TRACE_WPS_TLAS;
    if (stats) { if (inst) hipLaunchKernelGGL((k_trace_shadow<true, true>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, q, cnt, spill, overflow, stat_out, refill);
                 else hipLaunchKernelGGL((k_trace_shadow<true, false>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, q, cnt, spill, overflow, stat_out, refill); }
    else { if (inst) hipLaunchKernelGGL((k_trace_shadow<false, true>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, q, cnt, spill, overflow, stat_out, refill);
           else hipLaunchKernelGGL((k_trace_shadow<false, false>), dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, q, cnt, spill, overflow, stat_out, refill); }
}
void launch_trace_probe(hipStream_t s, int grid, const SceneView& sc, const float* rays, uint32_t n, int any_hit, uint32_t* head, uint32_t* out_ids, float* out_tuv,
                        uint32_t* spill, uint32_t* overflow, uint32_t refill) {
    grid = grid / TRACE_WPS * TRACE_WPS_TLAS;
    if (any_hit) hipLaunchKernelGGL(k_trace_probe<true>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, rays, n, head, out_ids, out_tuv, spill, overflow, refill);
    else hipLaunchKernelGGL(k_trace_probe<false>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, rays, n, head, out_ids, out_tuv, spill, overflow, refill);
}
int trace_blocks_per_cu() { return TRACE_WPS; }
size_t trace_spill_words(int grid) { return (size_t)grid * TRACE_BLOCK * STACK_SPILL * 2; }

}  // namespace msne
