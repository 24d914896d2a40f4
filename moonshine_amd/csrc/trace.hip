// trace.hip — k_trace_closest / k_trace_shadow: software replacement for the reference's
// TraceRay calls (shaders/hrtsystem/intersection.hlsl:18-22 closest hit, :33-46 any hit; payload
// written by closesthit/miss/shadowmiss in main.hlsl:102-118) over the driver-built TLAS/BLAS
// (engine/hrtsystem/Accel.zig:94-184,484).
//
// Design (gfx950): persistent waves pull 64 rays at a time from the compacted ray queue with one
// atomicAdd per wave; each lane walks the two-level 8-wide quantized BVH (80-B nodes, 48-B triangles)
// with a short per-lane stack in LDS ([entry][thread] layout → conflict-free ds_read/ds_write_b32),
// spilling to HBM only beyond STACK_LDS entries.  No MFMA: this is pointer chasing + 3-vector math.
// Box tests use fmaf and a relative slack (they only gate which triangles are tested); the triangle
// test is the watertight Woop–Benthin–Wald test evaluated op-for-op like the test oracle, and equal-t
// ties resolve to the smallest (instance, geometry, primitive), so results do not depend on BVH shape.
#include "msne_device.h"

namespace msne {

constexpr int TRACE_BLOCK = 256;
constexpr int STACK_LDS = 24;
constexpr int STACK_SPILL = 232;          // total depth 256 entries per lane
constexpr uint32_t ENT_SENTINEL = 0xFFFFFFFFu;
constexpr uint32_t ENT_KIND_TRI = 1u << 30;
constexpr uint32_t ENT_KIND_INST = 2u << 30;

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

// watertight ray/triangle test (Woop, Benthin, Wald 2013), no culling; (u,v) = weights of vertices 1,2
__device__ __forceinline__ bool tri_intersect(f3 o, const RayK& k, f3 v0, f3 v1, f3 v2, float& t, float& u, float& v) {
    const f3 A = sub(v0, o), B = sub(v1, o), C = sub(v2, o);
    const float Akz = idx3(A, k.kz), Bkz = idx3(B, k.kz), Ckz = idx3(C, k.kz);
    const float Ax = idx3(A, k.kx) - k.Sx * Akz, Ay = idx3(A, k.ky) - k.Sy * Akz;
    const float Bx = idx3(B, k.kx) - k.Sx * Bkz, By = idx3(B, k.ky) - k.Sy * Bkz;
    const float Cx = idx3(C, k.kx) - k.Sx * Ckz, Cy = idx3(C, k.ky) - k.Sy * Ckz;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx; U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx; V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax; W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    const float det = U + V + W;
    if (det == 0.0f) return false;
    const float Az = k.Sz * Akz, Bz = k.Sz * Bkz, Cz = k.Sz * Ckz;
    const float T = U * Az + V * Bz + W * Cz;
    const float rcp = 1.0f / det;
    const float tt = T * rcp;
    if (!(tt > 0.0f)) return false;
    t = tt; u = V * rcp; v = W * rcp;
    return true;
}

__device__ __forceinline__ float safe_inv(float d) { return absf(d) < 1e-30f ? (d < 0.0f ? -1e30f : 1e30f) : 1.0f / d; }

struct Hit { uint32_t inst, geo, prim; float t, u, v; };

struct TraceStats { unsigned long long node_visits, tri_tests; };

template <bool ANY_HIT, bool STATS>
__device__ __forceinline__ bool traverse(const SceneView& sc, f3 o_w, f3 d_w, float tmax, Hit& best,
                                         uint32_t* lds_stack, uint32_t* spill, uint32_t spill_stride, uint32_t* overflow,
                                         TraceStats* stats) {
    best.inst = MAX_UINT; best.geo = 0; best.prim = 0; best.t = tmax; best.u = 0.0f; best.v = 0.0f;
    if (sc.tlas_root == MAX_UINT) return false;
    const uint32_t tid = threadIdx.x;
    int sp = 0;
    auto push = [&](uint32_t e) {
        if (sp < STACK_LDS) lds_stack[sp * TRACE_BLOCK + tid] = e;
        else if (sp < STACK_LDS + STACK_SPILL) spill[(size_t)(sp - STACK_LDS) * spill_stride] = e;
        else { *overflow = 1u; return; }
        sp++;
    };
    auto pop = [&]() -> uint32_t {
        sp--;
        return sp < STACK_LDS ? lds_stack[sp * TRACE_BLOCK + tid] : spill[(size_t)(sp - STACK_LDS) * spill_stride];
    };

    f3 o = o_w, d = d_w;
    f3 id = F3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
    RayK rk = rayk_make(d);
    bool in_blas = false;
    uint32_t cur_inst = 0;
    uint32_t cur = sc.tlas_root;     // entry being processed
    bool have = true;
    unsigned long long nv = 0, nt = 0;

    for (;;) {
        if (!have) {
            if (sp == 0) break;
            cur = pop();
        }
        have = false;
        if (cur == ENT_SENTINEL) {           // leaving an instance: back to world space
            o = o_w; d = d_w;
            id = F3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
            rk = rayk_make(d);
            in_blas = false;
            continue;
        }
        const uint32_t kind = cur >> 30;
        if (kind == 0) {
            // ---- internal node: 5 x 16-B loads ----
            const uint4* np = reinterpret_cast<const uint4*>(sc.nodes + cur);
            const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3], w4 = np[4];
            if (STATS) nv++;
            const float nox = u2f(w0.x), noy = u2f(w0.y), noz = u2f(w0.z);
            const uint32_t ex = w0.w & 0xff, ey = (w0.w >> 8) & 0xff, ez = (w0.w >> 16) & 0xff, imask = w0.w >> 24;
            const uint32_t child_base = w1.x, item_base = w1.y;
            const uint32_t meta_lo = w1.z, meta_hi = w1.w;
            const float ax = u2f(ex << 23) * id.x, ay = u2f(ey << 23) * id.y, az = u2f(ez << 23) * id.z;
            const float bx = (nox - o.x) * id.x, by = (noy - o.y) * id.y, bz = (noz - o.z) * id.z;
            // byte planes: qlo[0] = w2.xy, qlo[1] = w2.zw, qlo[2] = w3.xy, qhi[0] = w3.zw, qhi[1] = w4.xy, qhi[2] = w4.zw
            const uint32_t qw[12] = { w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, w4.z, w4.w };
            float tn[8]; uint32_t hitmask = 0;
            const float tlimit = best.t;
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
                const bool valid = ((imask >> i) & 1u) || m != 0xffu;
                tn[i] = n;
                if (valid && n <= f * 1.00001f) hitmask |= 1u << i;
            }
            if (hitmask) {
                // nearest hit child is processed next; the others go on the stack
                int nearest = -1; float nt_ = 3.0e38f;
#pragma unroll
                for (int i = 0; i < 8; i++) if (((hitmask >> i) & 1u) && tn[i] < nt_) { nt_ = tn[i]; nearest = i; }
                uint32_t next_entry = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (!((hitmask >> i) & 1u)) continue;
                    uint32_t e;
                    if ((imask >> i) & 1u) e = child_base + __popc(imask & ((1u << i) - 1u));
                    else {
                        const uint32_t m = ((i < 4 ? meta_lo : meta_hi) >> ((i & 3) * 8)) & 0xff;
                        const uint32_t first = item_base + (m & 31u);
                        e = in_blas ? (ENT_KIND_TRI | ((m >> 5) << 28) | first) : (ENT_KIND_INST | first);
                    }
                    if (i == nearest) next_entry = e; else push(e);
                }
                cur = next_entry; have = true;
            }
        } else if (kind == 1) {
            // ---- triangle leaf ----
            const uint32_t count = ((cur >> 28) & 3u) + 1u, first = cur & 0x0FFFFFFFu;
            for (uint32_t q = 0; q < count; q++) {
                const uint4* tp = reinterpret_cast<const uint4*>(sc.tris + first + q);
                const uint4 a = tp[0], b = tp[1], c = tp[2];
                if (STATS) nt++;
                float t, u, v;
                if (!tri_intersect(o, rk, F3(u2f(a.x), u2f(a.y), u2f(a.z)), F3(u2f(a.w), u2f(b.x), u2f(b.y)), F3(u2f(b.z), u2f(b.w), u2f(c.x)), t, u, v)) continue;
                if (ANY_HIT) {
                    if (t < best.t) { if (STATS) { stats->node_visits = nv; stats->tri_tests = nt; } return true; }
                    continue;
                }
                bool closer = t < best.t;
                if (!closer && t == best.t && best.inst != MAX_UINT)
                    closer = cur_inst < best.inst || (cur_inst == best.inst && (c.y < best.geo || (c.y == best.geo && c.z < best.prim)));
                if (closer) { best.t = t; best.u = u; best.v = v; best.inst = cur_inst; best.geo = c.y; best.prim = c.z; }
            }
        } else {
            // ---- instance leaf: enter the BLAS in instance space (t is preserved: d is not renormalised) ----
            const uint32_t item = cur & 0x3FFFFFFFu;
            const uint32_t ii = sc.tlas_items[item];
            const InstanceRec* ir = sc.instances + ii;
            const uint32_t root = ir->blas_root, flags = ir->flags;
            if (!(flags & 1u) || root == MAX_UINT) continue;
            const float4* mp = reinterpret_cast<const float4*>(&ir->world_to_instance);
            const float4 r0 = mp[0], r1 = mp[1], r2 = mp[2];
            m34 M;
            M.m[0][0] = r0.x; M.m[0][1] = r0.y; M.m[0][2] = r0.z; M.m[0][3] = r0.w;
            M.m[1][0] = r1.x; M.m[1][1] = r1.y; M.m[1][2] = r1.z; M.m[1][3] = r1.w;
            M.m[2][0] = r2.x; M.m[2][1] = r2.y; M.m[2][2] = r2.z; M.m[2][3] = r2.w;
            o = m34_mul_point(M, o_w);
            d = m34_mul_vec(M, d_w);
            id = F3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
            rk = rayk_make(d);
            in_blas = true; cur_inst = ii;
            push(ENT_SENTINEL);
            cur = root; have = true;
        }
    }
    if (STATS) { stats->node_visits = nv; stats->tri_tests = nt; }
    return best.inst != MAX_UINT;
}

// ---------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_closest(SceneView sc, PathState st, HitBuf hits, Counters* cnt,
                                                                uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out) {
    __shared__ uint32_t lds_stack[STACK_LDS * TRACE_BLOCK];
    const uint32_t n = cnt->n_cur;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * TRACE_BLOCK + threadIdx.x;
    const uint32_t spill_stride = gridDim.x * TRACE_BLOCK;
    unsigned long long nv = 0, nt = 0;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&cnt->head_closest, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) break;
        const uint32_t i = base + lane;
        if (i < n) {
            Hit h; h.inst = MAX_UINT; h.geo = 0; h.prim = 0; h.u = 0.0f; h.v = 0.0f;
            if (!(st.flags[i] & PATH_FLAG_ZOMBIE)) {
                const f3 o = F3(st.ox[i], st.oy[i], st.oz[i]), d = F3(st.dx[i], st.dy[i], st.dz[i]);
                TraceStats ts{ 0, 0 };
                traverse<false, STATS>(sc, o, d, INFINITY_F, h, lds_stack, spill + gtid, spill_stride, overflow, &ts);
                if (STATS) { nv += ts.node_visits; nt += ts.tri_tests; }
            }
            hits.inst[i] = h.inst; hits.geo[i] = h.geo; hits.prim[i] = h.prim; hits.u[i] = h.u; hits.v[i] = h.v;
        }
    }
    if (STATS) { atomicAdd(&stat_out[0], nv); atomicAdd(&stat_out[1], nt); }
}

template <bool STATS>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_shadow(SceneView sc, ShadowQueue q, PathState next, Counters* cnt,
                                                               uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out) {
    __shared__ uint32_t lds_stack[STACK_LDS * TRACE_BLOCK];
    const uint32_t n = cnt->n_shadow;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * TRACE_BLOCK + threadIdx.x;
    const uint32_t spill_stride = gridDim.x * TRACE_BLOCK;
    unsigned long long nv = 0, nt = 0;
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&cnt->head_shadow, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= n) break;
        const uint32_t i = base + lane;
        if (i < n) {
            const f3 o = F3(q.ox[i], q.oy[i], q.oz[i]), d = F3(q.dx[i], q.dy[i], q.dz[i]);
            Hit h; TraceStats ts{ 0, 0 };
            const bool occluded = traverse<true, STATS>(sc, o, d, q.tmax[i], h, lds_stack, spill + gtid, spill_stride, overflow, &ts);
            if (STATS) { nv += ts.node_visits; nt += ts.tri_tests; }
            if (occluded) {   // ShadowIntersection::hit → lightSample.pdf = 0 (light.hlsl:75-77,154-156): the pending contribution vanishes
                const uint32_t tg = q.target[i], j = tg >> 1;
                if (tg & 1u) { next.p1x[j] = 0.0f; next.p1y[j] = 0.0f; next.p1z[j] = 0.0f; }
                else { next.p0x[j] = 0.0f; next.p0y[j] = 0.0f; next.p0z[j] = 0.0f; }
            }
        }
    }
    if (STATS) { atomicAdd(&stat_out[2], nv); atomicAdd(&stat_out[3], nt); }
}

// probe kernel for tests: arbitrary rays → hit records (closest) or occlusion flags (any)
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_probe(SceneView sc, const float* rays /*7 per ray: o,d,tmax*/, uint32_t n, int any_hit,
                                                              uint32_t* out_ids /*4 per ray: hit,inst,geo,prim*/, float* out_tuv /*3 per ray*/,
                                                              uint32_t* spill, uint32_t* overflow) {
    __shared__ uint32_t lds_stack[STACK_LDS * TRACE_BLOCK];
    const uint32_t gtid = blockIdx.x * TRACE_BLOCK + threadIdx.x;
    const uint32_t spill_stride = gridDim.x * TRACE_BLOCK;
    for (uint32_t i = gtid; i < n; i += spill_stride) {
        const float* r = rays + 7 * (size_t)i;
        Hit h; TraceStats ts{ 0, 0 };
        bool hit;
        if (any_hit) hit = traverse<true, false>(sc, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], h, lds_stack, spill + gtid, spill_stride, overflow, &ts);
        else hit = traverse<false, false>(sc, F3(r[0], r[1], r[2]), F3(r[3], r[4], r[5]), r[6], h, lds_stack, spill + gtid, spill_stride, overflow, &ts);
        out_ids[4 * i] = hit ? 1u : 0u; out_ids[4 * i + 1] = h.inst; out_ids[4 * i + 2] = h.geo; out_ids[4 * i + 3] = h.prim;
        out_tuv[3 * i] = h.t; out_tuv[3 * i + 1] = h.u; out_tuv[3 * i + 2] = h.v;
    }
}

// ---------------- host launch wrappers ----------------
void launch_trace_closest(hipStream_t s, int grid, bool stats, const SceneView& sc, const PathState& st, const HitBuf& hits, Counters* cnt,
                          uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out) {
    if (stats) hipLaunchKernelGGL(k_trace_closest<true>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, st, hits, cnt, spill, overflow, stat_out);
    else hipLaunchKernelGGL(k_trace_closest<false>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, st, hits, cnt, spill, overflow, stat_out);
}
void launch_trace_shadow(hipStream_t s, int grid, bool stats, const SceneView& sc, const ShadowQueue& q, const PathState& next, Counters* cnt,
                         uint32_t* spill, uint32_t* overflow, unsigned long long* stat_out) {
    if (stats) hipLaunchKernelGGL(k_trace_shadow<true>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, q, next, cnt, spill, overflow, stat_out);
    else hipLaunchKernelGGL(k_trace_shadow<false>, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, q, next, cnt, spill, overflow, stat_out);
}
void launch_trace_probe(hipStream_t s, int grid, const SceneView& sc, const float* rays, uint32_t n, int any_hit, uint32_t* out_ids, float* out_tuv,
                        uint32_t* spill, uint32_t* overflow) {
    hipLaunchKernelGGL(k_trace_probe, dim3(grid), dim3(TRACE_BLOCK), 0, s, sc, rays, n, any_hit, out_ids, out_tuv, spill, overflow);
}
size_t trace_spill_words(int grid) { return (size_t)grid * TRACE_BLOCK * STACK_SPILL; }

}  // namespace msne
