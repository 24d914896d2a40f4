// integrator.hip — wavefront form of the reference's raygen megakernel:
//   k_raygen  = raygen prologue: Rng::fromSeed, dispatchUV, Camera::generateRay       (main.hlsl:54-59,83-89; camera.hlsl:14-42)
//   k_shade   = one iteration of PathTracingIntegrator::incomingRadiance              (integrator.hlsl:79-166) + the miss epilogue (:168-181)
//   k_film    = storeColor                                                            (main.hlsl:43-51,94)
//   k_account = ray / sample statistics of a finished batch (no reference equivalent)
// Light samples are generated in k_shade and stored, as if unoccluded, next to their shadow rays in the shadow queue;
// k_trace_shadow zeroes the contribution of an occluded ray; the next k_shade adds a path's contributions to its radiance
// in the reference's order (env samples, then mesh samples — integrator.hlsl:139-151).  Any number of samples per bounce.
#include "shade.h"

namespace msne {

#ifndef SHADE_BLOCK_THREADS
#define SHADE_BLOCK_THREADS 256
#endif
constexpr int SHADE_BLOCK = SHADE_BLOCK_THREADS;   // one wave per SIMD per workgroup; k_shade runs four workgroups per CU (128 registers, 39 KB of LDS each)

__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (!pred) return 0;
    const uint32_t lane = __lane_id();
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// pixel owned by local pixel index p (tile-major order inside the shard); false if outside the image
__device__ __forceinline__ bool shard_pixel(const ShardView& sh, uint32_t p, uint32_t& x, uint32_t& y) {
    const uint32_t tpix = sh.tile_size * sh.tile_size;
    const uint32_t k = p / tpix, w = p % tpix;
    const uint32_t t = sh.shard_index + k * sh.shard_count;
    const uint32_t tx = t % sh.tiles_x, ty = t / sh.tiles_x;
    x = tx * sh.tile_size + w % sh.tile_size;
    y = ty * sh.tile_size + w / sh.tile_size;
    return x < sh.width && y < sh.height;
}

__global__ __launch_bounds__(SHADE_BLOCK) void k_raygen(ShardView sh, CameraConsts cam, PipelineOpts opts, uint32_t sample_base, uint32_t s_count,
                                                          PathState st, BounceCounters* cnt) {
    if (blockIdx.x == 0 && threadIdx.x < QUEUE_SUBS) cnt->sub[threadIdx.x].n_paths = queue_dense_len(s_count * sh.pixels, threadIdx.x);   // bounce 0's queue: dense (msne_device.h)
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->zombies = s_count * (sh.pixels - sh.valid_pixels);
    // queue index = slot (no compaction, no atomics: one device-scope counter would cap this kernel at ~88 waves/us);
    // the few slots of edge tiles that fall outside the image are flagged and dropped by the first k_shade
    const uint32_t total = s_count * sh.pixels;
    for (uint32_t slot = blockIdx.x * SHADE_BLOCK + threadIdx.x; slot < total; slot += gridDim.x * SHADE_BLOCK) {
        uint32_t x = 0, y = 0;
        const uint32_t i = slot;
        if (!shard_pixel(sh, slot % sh.pixels, x, y)) { nt_store(&st.ro[i], make_float4(0.0f, 0.0f, 0.0f, u2f(PATH_FLAG_ZOMBIE | PATH_FLAG_MASKED))); continue; }
        const uint32_t s_local = slot / sh.pixels;
        uint32_t rng = rng_seed(sample_base + s_local, x, y);                       // main.hlsl:85
        f2 r1; r1.x = rng_float(rng); r1.y = rng_float(rng);
        const f2 g = square_to_gaussian(r1);                                        // dispatchUV main.hlsl:54-59
        const f2 center = F2(0.5f + 0.5f * g.x, 0.5f + 0.5f * g.y);
        f2 uv = F2(((float)x + center.x) / (float)sh.width, ((float)y + center.y) / (float)sh.height);
        if (opts.flip_image) uv.y = 1.0f - uv.y;
        f2 r2; r2.x = rng_float(rng); r2.y = rng_float(rng);
        f3 O, D;
        camera_generate_ray(cam, uv, r2, O, D);
        // a camera path starts with throughput 1, radiance 0 and its queue index as its slot: only the ray and the RNG state (in rd.w, which nothing else uses) are
        // written — 32 B per path instead of 80; the first k_shade (first_pass) fills in the rest itself
        nt_store(&st.ro[i], make_float4(O.x, O.y, O.z, u2f(0u))); nt_store(&st.rd[i], make_float4(D.x, D.y, D.z, u2f(rng)));
    }
}

// one record per alias-table entry (+ one for the zero entry): the loads of MeshAttributes::lookupAndInterpolate (world.hlsl:114-158), done once
__global__ void k_light_tris(SceneView sc, uint32_t indexed_attributes, uint32_t n_instances, LightTri* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > sc.alias_count) return;
    AliasEntry en; en.alias = 0; en.select = 0.0f; en.instance = 0; en.geometry = 0; en.primitive = 0;
    if (i < sc.alias_count) en = sc.alias[1 + i];
    LightTri r;
    r.p0x = r.p0y = r.p0z = r.p1x = r.p1y = r.p1z = r.p2x = r.p2y = r.p2z = 0.0f;
    r.t0x = 0.0f; r.t0y = 0.0f; r.t1x = 1.0f; r.t1y = 0.0f; r.t2x = 1.0f; r.t2y = 1.0f; r.material = 0;
    r.nx = r.ny = r.nz = 0.0f; r.instance = en.instance;
    for (int k = 0; k < 12; k++) r.to_world[k] = 0.0f;
    r.emissive.offset = 0; r.emissive.w = 1; r.emissive.h = 1; r.emissive.format = TEX_RGBA32F; r.emissive.first = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (en.instance < n_instances) {
        const GeometryRec g = sc.geometries[sc.instances[en.instance].geo_offset + en.geometry];
        const MeshRec mesh = sc.meshes[g.mesh];
        const uint32_t i0 = mesh.indices[3 * (size_t)en.primitive], i1 = mesh.indices[3 * (size_t)en.primitive + 1], i2 = mesh.indices[3 * (size_t)en.primitive + 2];
        const f3 p0 = ld3(mesh.positions, i0), p1 = ld3(mesh.positions, i1), p2 = ld3(mesh.positions, i2);
        r.p0x = p0.x; r.p0y = p0.y; r.p0z = p0.z; r.p1x = p1.x; r.p1y = p1.y; r.p1z = p1.z; r.p2x = p2.x; r.p2y = p2.y; r.p2z = p2.z;
        if (mesh.texcoords) {
            const uint32_t a0 = indexed_attributes ? i0 : en.primitive * 3 + 0, a1 = indexed_attributes ? i1 : en.primitive * 3 + 1, a2 = indexed_attributes ? i2 : en.primitive * 3 + 2;
            const f2 t0 = ld2(mesh.texcoords, a0), t1 = ld2(mesh.texcoords, a1), t2 = ld2(mesh.texcoords, a2);
            r.t0x = t0.x; r.t0y = t0.y; r.t1x = t1.x; r.t1y = t1.y; r.t2x = t2.x; r.t2y = t2.y;
        }
        r.material = g.material;
        const InstanceRec* inst = sc.instances + en.instance;
        const m34 tw = inst->transform, tm = inst->world_to_instance;
        for (int a = 0; a < 3; a++) for (int b = 0; b < 4; b++) r.to_world[4 * a + b] = tw.m[a][b];
        const f3 n = normalize(m34_mul_transposed(tm, normalize(cross(sub(p0, p2), sub(p1, p2)))));   // triangleFrame.n after inWorld(): world.hlsl:145,171 + reflection_frame.hlsl:24
        r.nx = n.x; r.ny = n.y; r.nz = n.z;
        r.emissive = sc.textures[sc.materials[g.material].emissive];
    }
    out[i] = r;
}

__device__ __forceinline__ float power_heuristic(uint32_t numf, float fPdf, uint32_t numg, float gPdf) {   // integrator.hlsl:10-16
    const float f = (float)numf * fPdf, g = (float)numg * gPdf;
    const float f2_ = f * f;
    return f2_ / (f2_ + g * g);
}
// estimateDirectMISLight integrator.hlsl:20-35, light already sampled and assumed unoccluded
__device__ __forceinline__ f3 estimate_direct_mis(const Frame& frame, const LSample& ls, const Mat& mat, f3 woFs, uint32_t samplesTaken) {
    if (ls.pdf > 0.0f) {
        const f3 wiFs = frame_world_to_frame(frame, ls.dirWs);
        const float scatteringPdf = material_pdf(mat, wiFs, woFs);
        if (scatteringPdf > 0.0f) {
            const f3 brdf = material_eval(mat, wiFs, woFs);
            const float weight = power_heuristic(samplesTaken, ls.pdf, 1, scatteringPdf);
            const float ac = absf(wiFs.z);
            return F3(ls.radiance.x * brdf.x * ac * weight / ls.pdf, ls.radiance.y * brdf.y * ac * weight / ls.pdf, ls.radiance.z * brdf.z * ac * weight / ls.pdf);
        }
    }
    return F3(0.0f, 0.0f, 0.0f);
}

#ifndef MSNE_ENV_TOP_LDS
#define MSNE_ENV_TOP_LDS 1   // (0: the descent reads every level from memory, as before round 6 — tools/variant_rates.py)
#endif
#ifndef SHADE_WPS
#define SHADE_WPS 4          // resident waves per SIMD the TEXTURED k_shade is register-allocated for: 128 registers with the descriptors fetched where they are used (19 spilled;
                             // stand-in with 64^2 / 1024^2 textures: shade 55.9 -> 54.0 / 60.7 -> 59.5 ms per 64-step batch against 3 waves at 166 registers; round 3, with the
                             // descriptors fetched up front, had measured 4 waves slower)
#endif
// SPEC: which paths an instantiation shades — 0: all of them (the shipped kernel); 1: finished paths and misses (no surface code at all); 2: hits on glass and
// mirrors (no light samples, no PBR); 3: Lambert hits; 4: StandardPBR hits.  The specialised ones skip every other path of the queue: launched one after the
// other they shade a bounce between them, bit-identically ($MSNE_SHADE_SPEC=1; measured in profiles/r04_shade_specialised.txt).
// TEX: false when every texture of the scene is 1x1 (constant material parameters): no sampler code, fewer registers (shade.h tex_sample_desc)
#ifndef SHADE_WPS_CONST
#define SHADE_WPS_CONST 4    // 128 registers, 8 spilled: S1 k_shade 26.7 -> 23.0 ms per 64-step batch (5904 -> 6197 Mrays/s), S2 39.0 -> 33.0 (3579 -> 3707); at 3 waves (141 registers) 26.1 / 38.2
#endif
constexpr uint32_t shade_spec_wps(int spec, bool tex) { return spec == 1 ? 4u : (tex ? (uint32_t)SHADE_WPS : (uint32_t)SHADE_WPS_CONST); }   // (the workgroup's 39 KB of LDS allow four waves per SIMD at most)
// TRUNC (measurement only, profiles/r05_shade_td.txt): 1 = the kernel ends after the path-state loads and their LDS staging, 2 = after the hit / geometry / material chain and
// the sort, 3 = after the hit's attributes, frames and material parameters are fetched, 4 = after the queue slots are reserved (one atomic per workgroup, on the counters'
// padding), 5 = after the light samples, 6 = everything but the stores (nothing is ever written); launched in front of the real kernel under $MSNE_SHADE_TRUNC
template <int SPEC, bool TEX, int TRUNC = 0>
__global__ __launch_bounds__(SHADE_BLOCK, shade_spec_wps(SPEC, TEX)) void k_shade(SceneView sc, PipelineOpts opts, PathState cur, HitBuf hits, PathState nxt, ShadowQueue shq,
                                                         const float4* c_prev /* light-sample contributions of the previous bounce */,
                                                         float4* lbuf, BounceCounters* cnt /* [0]: this bounce, [1]: the next */, uint32_t first_pass /* the queue is k_raygen's */) {
    const QueueDims qd = queue_dims<false>(&cnt[0]);   // this bounce's path queue: its extent, and which of its entries hold a path (msne_device.h)
    const uint32_t n = qd.extent;
    const uint32_t max_bounces = opts.max_bounces, env_n = opts.env_samples, mesh_n = opts.mesh_samples;
    const bool have_lights = !(sc.alias_count == 0 || sc.alias_sum == 0.0f);                  // MeshLights::sample returns pdf 0 without them (light.hlsl:131)
    const uint32_t n_nee = env_n + (have_lights ? mesh_n : 0u);                               // shadow-queue entries per path that samples lights
    __shared__ unsigned long long s_cnt[SHADE_BLOCK / 64];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_stride;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const uint32_t n_pad = (n + (SHADE_BLOCK - 1u)) & ~(SHADE_BLOCK - 1u);   // trip count uniform per workgroup (it synchronises below)
    // Paths are shaded in an order sorted by what they will execute (finalise / miss / material type), 256 at a time: the
    // reference's megakernel diverges per pixel, here a wave runs one BSDF.  Results are per sample slot, so the order
    // inside a workgroup is free.
    constexpr uint32_t CAT_NONE = 6u, NCAT = 6u;   // 0 zombie, 1 miss, 2 + material type (MAT_GLASS .. MAT_PBR)
    __shared__ uint32_t s_cat[NCAT * (SHADE_BLOCK / 64) + 1];
    __shared__ uint32_t s_perm[SHADE_BLOCK];
    __shared__ uint4 s_hit[SHADE_BLOCK], s_geo[SHADE_BLOCK], s_mat[2][SHADE_BLOCK];   // what the sort already fetched: hit, geometry and material records
    // the path state travels through LDS too: every thread loads the record of ITS queue entry (five coalesced wave-wide loads) and the
    // thread that shades the path after the sort picks it up there, instead of gathering 16-B pieces from a permuted index
    __shared__ float4 s_ro[SHADE_BLOCK], s_rd[SHADE_BLOCK], s_tp[SHADE_BLOCK], s_lr[SHADE_BLOCK];
    __shared__ uint2 s_sq[SHADE_BLOCK];
    __shared__ float4 s_envtop[ENV_TOP_QUADS];   // the top of the environment's sampling pyramid (shade.h env_top_stage); read after the first barrier of the loop below
    if (MSNE_ENV_TOP_LDS && SPEC != 1 && SPEC != 2 && env_n != 0u) env_top_stage(sc.env, s_envtop, threadIdx.x, SHADE_BLOCK);
    uint32_t acc = 0;   // TRUNC: what the truncated kernel folds its loads into
    for (uint32_t base_i = blockIdx.x * SHADE_BLOCK; base_i < n_pad; base_i += gridDim.x * SHADE_BLOCK) {
        uint32_t cat = CAT_NONE;
        {
            const uint32_t i0 = base_i + threadIdx.x;
            if (queue_live<false>(qd, i0)) {
                const float4 ro_own = nt_load(&cur.ro[i0]);
                const float4 rd_own = nt_load(&cur.rd[i0]);
                s_ro[threadIdx.x] = ro_own; s_rd[threadIdx.x] = rd_own;
                if (first_pass) { s_tp[threadIdx.x] = make_float4(1.0f, 1.0f, 1.0f, 0.0f); s_lr[threadIdx.x] = make_float4(0.0f, 0.0f, 0.0f, rd_own.w); s_sq[threadIdx.x] = make_uint2(i0, 0u); }
                const uint32_t fl = f2u(ro_own.w);
                if (!first_pass) {
                    s_tp[threadIdx.x] = nt_load(&cur.tp[i0]);
                    float4 lr_own = nt_load(&cur.lr[i0]); const uint2 sq_own = nt_load(&cur.sq[i0]);
                    // light samples of the previous bounce, in the reference's order (env samples, then mesh samples): each was stored unoccluded next to its shadow ray
                    // and zeroed by k_trace_shadow if the ray was blocked.  Added HERE, by the thread that owns the queue entry, as soon as the entry's flags and sample
                    // position are in: the fetches run next to the hit / geometry / material chain below instead of after the sort
                    if (fl & PATH_FLAG_NEE) {
                        const uint32_t pq = shq_pos(sq_own.y), pk = shq_sub(sq_own.y);   // position in its sub-queue | sub-queue (msne_device.h shq_pack)
                        const uint32_t ps = (fl >> PATH_STRIDE_SHIFT) & 0x1ffu;
                        for (uint32_t k = 0; k < n_nee; k++) { const float4 c = nt_load(&c_prev[queue_slot(pq + k * ps, pk)]); lr_own.x = lr_own.x + c.x; lr_own.y = lr_own.y + c.y; lr_own.z = lr_own.z + c.z; }
                    }
                    s_lr[threadIdx.x] = lr_own; s_sq[threadIdx.x] = sq_own;
                }
                if (TRUNC == 1) acc ^= fl ^ f2u(rd_own.x);
                if (TRUNC != 1 && !(fl & (PATH_FLAG_MASKED | PATH_FLAG_DEAD))) {
                    if (fl & PATH_FLAG_ZOMBIE) cat = 0u;
                    else {
                        const uint4 hr = nt_load(&hits.rec[i0]);
                        s_hit[threadIdx.x] = hr;
                        if (hr.x == MAX_UINT) cat = 1u;
                        else {
                            const uint32_t geo = sc.tris[hr.y].geo;
                            const GeometryRec g = sc.geometries[sc.instances[hr.x].geo_offset + geo];
                            const uint4* mp = reinterpret_cast<const uint4*>(sc.materials + g.material);
                            const uint4 m0 = mp[0], m1 = mp[1];
                            s_geo[threadIdx.x] = make_uint4(g.mesh, g.material, g.sampled, 0u);
                            s_mat[0][threadIdx.x] = m0; s_mat[1][threadIdx.x] = m1;
                            cat = 2u + (m0.z & 3u);
                        }
                    }
                }
                if (SPEC == 1 && cat > 1u) cat = CAT_NONE;
                if (SPEC == 2 && cat != 2u + MAT_GLASS && cat != 2u + MAT_MIRROR) cat = CAT_NONE;
                if (SPEC == 3 && cat != 2u + MAT_LAMBERT) cat = CAT_NONE;
                if (SPEC == 4 && cat != 2u + MAT_PBR) cat = CAT_NONE;
            }
            if (TRUNC == 1) continue;
            uint32_t rank = 0;
#pragma unroll
            for (uint32_t c = 0; c < NCAT; c++) {
                const unsigned long long m = __ballot(cat == c);
                if (cat == c) rank = (uint32_t)__popcll(m & lt);
                if (lane == 0) s_cat[c * (SHADE_BLOCK / 64) + wave] = (uint32_t)__popcll(m);
            }
            __syncthreads();
            if (threadIdx.x == 0) {   // exclusive scan over (category, wave); the total lands in the last entry
                uint32_t run = 0;
                for (uint32_t k = 0; k < NCAT * (SHADE_BLOCK / 64); k++) { const uint32_t v = s_cat[k]; s_cat[k] = run; run += v; }
                s_cat[NCAT * (SHADE_BLOCK / 64)] = run;
            }
            __syncthreads();
            if (cat != CAT_NONE) s_perm[s_cat[cat * (SHADE_BLOCK / 64) + wave] + rank] = threadIdx.x;
            __syncthreads();
        }
        const bool live = threadIdx.x < s_cat[NCAT * (SHADE_BLOCK / 64)];
        const uint32_t src = live ? s_perm[threadIdx.x] : 0u;   // the thread that classified this path
        if (TRUNC == 2) { acc ^= src; continue; }
        // ---- phase A: what the hit means for the path (pending light samples, miss epilogue, emission, termination) ----
        bool alive = false, nee = false, delta = false;
        f3 rayO = F3(0, 0, 0), rayD = F3(0, 0, 1), throughput = F3(0, 0, 0), L = F3(0, 0, 0), woSs = F3(0, 0, 1);
        float lastPdf = 0.0f;
        uint32_t rng = 0, slot = 0, flags = 0, bounceCount = 0;
        Attrs attrs; Frame shadingFrame; Mat material;
        attrs.position = F3(0, 0, 0); attrs.triangleFrame.n = F3(0, 0, 1); shadingFrame.n = F3(0, 0, 1); shadingFrame.s = F3(1, 0, 0); shadingFrame.t = F3(0, 1, 0);
        material.type = MAT_LAMBERT; material.color = F3(0, 0, 0); material.metalness = 0.0f; material.alpha = 0.0f; material.ior = 1.0f;
        if (live) {
            const float4 ro4 = s_ro[src];
            { const float4 rd4 = s_rd[src]; rayO = F3(ro4.x, ro4.y, ro4.z); rayD = F3(rd4.x, rd4.y, rd4.z); }
            const float4 tp4 = s_tp[src], lr4 = s_lr[src]; const uint2 sq2 = s_sq[src];
            throughput = F3(tp4.x, tp4.y, tp4.z); L = F3(lr4.x, lr4.y, lr4.z);
            lastPdf = tp4.w; rng = f2u(lr4.w); slot = sq2.x; flags = f2u(ro4.w);
            // (the previous bounce's light samples are already in L: added when the entry was staged)
            bounceCount = flags & 0xFFFFu;
            const bool isLastMaterialDelta = (flags & PATH_FLAG_DELTA) != 0;
            bool done = (flags & PATH_FLAG_ZOMBIE) != 0;
            const uint4 hrec = s_hit[src];
            const uint32_t hinst = hrec.x;
            if (!done && hinst == MAX_UINT) {
                // miss epilogue, integrator.hlsl:168-181
                if (env_n == 0 || bounceCount == 0 || isLastMaterialDelta) L = add(L, mul(throughput, env_incoming_radiance(sc.env, rayD)));
                else {
                    f3 rad; float pdf;
                    env_eval(sc.env, rayD, rad, pdf);
                    if (pdf > 0.0f) { const float weight = power_heuristic(1, lastPdf, env_n, pdf); L = add(L, scale(mul(throughput, rad), weight)); }
                }
                done = true;
            }
            if (SPEC != 1 && !done) {
                const uint32_t htri = hrec.y;
                const uint4 pg = s_geo[src];
                GeometryRec geometry; geometry.mesh = pg.x; geometry.material = pg.y; geometry.sampled = pg.z;
                MaterialRec mrec;
                { const uint4 m0 = s_mat[0][src], m1 = s_mat[1][src];
                  mrec.normal = m0.x; mrec.emissive = m0.y; mrec.type = m0.z; mrec.color = m0.w; mrec.metalness = m1.x; mrec.roughness = m1.y; mrec.ior = u2f(m1.z); mrec.pad = m1.w;
                  if (SPEC == 2) mrec.type = m0.z & 2u;          // (the category filter let only these types through: the compiler drops the other BSDFs)
                  if (SPEC == 3) mrec.type = MAT_LAMBERT;
                  if (SPEC == 4) mrec.type = MAT_PBR; }
                Frame textureFrame; f3 emissiveLight;
                if (TEX && SHADE_WPS >= 4) {   // at 128 registers five descriptors in flight are 40 registers too many: each is fetched where it is used
                    attrs = mesh_attributes_world(sc, opts.indexed_attributes != 0, hinst, 0u, 0u, F2(u2f(hrec.z), u2f(hrec.w)), geometry, htri, true);
                    textureFrame = get_texture_frame<TEX>(sc, sc.textures[mrec.normal], opts.two_component_normal_texture != 0, attrs.texcoord, attrs.frame);
                    const float4 em4 = tex_sample_desc<TEX>(sc, sc.textures[mrec.emissive], attrs.texcoord);
                    emissiveLight = F3(em4.x, em4.y, em4.z);
                    const bool has_color = mrec.type == MAT_PBR || mrec.type == MAT_LAMBERT;
                    material = material_load_desc<TEX>(sc, mrec, sc.textures[has_color ? mrec.color : mrec.normal], sc.textures[mrec.type == MAT_PBR ? mrec.metalness : mrec.normal], sc.textures[mrec.type == MAT_PBR ? mrec.roughness : mrec.normal], attrs.texcoord);
                } else {
                    // the material's texture descriptors, all at once and before the attribute fetch below hides their latency (unused slots read the normal map's again);
                    // constant-texture scenes only ever use the descriptors' inline texels
                    const TexDesc t_normal = sc.textures[mrec.normal], t_emissive = sc.textures[mrec.emissive];
                    const bool has_color = mrec.type == MAT_PBR || mrec.type == MAT_LAMBERT;
                    const TexDesc t_color = sc.textures[has_color ? mrec.color : mrec.normal];
                    const TexDesc t_metal = sc.textures[mrec.type == MAT_PBR ? mrec.metalness : mrec.normal], t_rough = sc.textures[mrec.type == MAT_PBR ? mrec.roughness : mrec.normal];
                    attrs = mesh_attributes_world(sc, opts.indexed_attributes != 0, hinst, 0u, 0u, F2(u2f(hrec.z), u2f(hrec.w)), geometry, htri, true);
                    textureFrame = get_texture_frame<TEX>(sc, t_normal, opts.two_component_normal_texture != 0, attrs.texcoord, attrs.frame);
                    const float4 em4 = tex_sample_desc<TEX>(sc, t_emissive, attrs.texcoord);
                    emissiveLight = F3(em4.x, em4.y, em4.z);
                    material = material_load_desc<TEX>(sc, mrec, t_color, t_metal, t_rough, attrs.texcoord);
                }

                if (TRUNC == 3) acc ^= f2u(attrs.position.x) ^ f2u(attrs.texcoord.x) ^ f2u(attrs.frame.n.x) ^ f2u(attrs.triangleFrame.n.y) ^ f2u(textureFrame.n.z) ^ f2u(emissiveLight.x) ^ f2u(material.color.x) ^ f2u(material.alpha) ^ f2u(material.metalness);
                if (TRUNC != 3) {
                const f3 woWs = neg(rayD);
                const bool frontfacing = dot(attrs.triangleFrame.n, woWs) > 0.0f;
                if ((frontfacing && dot(woWs, textureFrame.n) > 0.0f) || (!frontfacing && -dot(woWs, textureFrame.n) > 0.0f)) shadingFrame = textureFrame;
                else if ((frontfacing && dot(woWs, attrs.frame.n) > 0.0f) || (!frontfacing && -dot(woWs, attrs.frame.n) > 0.0f)) shadingFrame = attrs.frame;
                else shadingFrame = attrs.triangleFrame;
                woSs = frame_world_to_frame(shadingFrame, woWs);

                // emission, integrator.hlsl:108-124
                const bool geo_sampled = (geometry.sampled & GEO_SAMPLED) != 0;
                if (mesh_n == 0 || bounceCount == 0 || !geo_sampled || isLastMaterialDelta) {
                    if (dot(woWs, attrs.triangleFrame.n) > 0.0f) L = add(L, mul(throughput, emissiveLight));
                } else if (geo_sampled) {
                    const float sum = sc.alias_sum;
                    const float lightPdf = area_to_solid_angle(attrs.position, rayO, rayD, attrs.triangleFrame.n) / sum;
                    if (lightPdf > 0.0f) { const float weight = power_heuristic(1, lastPdf, mesh_n, lightPdf); L = add(L, scale(mul(throughput, emissiveLight), weight)); }
                }
                // termination, integrator.hlsl:126-135
                if (bounceCount >= max_bounces + 1) done = true;
                else if (bounceCount > 3) {
                    const float pSurvive = minf(0.95f, luminance(throughput));
                    if (rng_float(rng) > pSurvive) done = true;
                    else throughput = divs(throughput, pSurvive);
                }
                if (!done) { alive = true; delta = SPEC == 2 ? true : (SPEC >= 3 ? false : material_is_delta(material)); nee = !delta && n_nee != 0u; }
                }
            }
            if (TRUNC != 0) acc ^= f2u(L.x) ^ f2u(L.y) ^ f2u(L.z);
            else if (done) nt_store(&lbuf[slot], make_float4(L.x, L.y, L.z, 0.0f));
        }
        if (TRUNC == 3) continue;
        // ---- phase B: queue slots.  One 64-bit atomic per WORKGROUP reserves both ranges (low word: one next-path entry per
        // surviving path, high word: n_nee shadow-ray entries per path that samples lights) — per-wave atomics on one counter
        // cap the kernel at ~88 waves/us (MI355X_MICROARCH.md "dequeue").  Reserving before the samples exist means they are
        // written straight to the queue instead of waiting in registers; a sample with pdf 0 leaves an unused entry. ----
        const unsigned long long ma = __ballot(alive), mn = __ballot(nee);
        if (lane == 0) s_cnt[wave] = (unsigned long long)__popcll(ma) | ((unsigned long long)__popcll(mn) << 32);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long tot = 0;
            for (int k = 0; k < SHADE_BLOCK / 64; k++) tot += s_cnt[k];
            const uint32_t np = (uint32_t)(tot >> 32);   // paths of this workgroup that sample lights: sample k of all of them is stored together
            s_stride = np;                               // (keeps env rays with env rays and light rays with light rays in the shadow queue)
            // the survivors of chunk c go to sub-queue c % QUEUE_SUBS of the next bounce's queues (msne_device.h): eight heads instead of one
            QueueHead* const head = &cnt[1].sub[(base_i >> QUEUE_TILE_SHIFT) & (QUEUE_SUBS - 1u)];
            if (TRUNC == 7) s_base = tot;   // (7 = 4 without the atomic: what the three barriers and the LDS traffic cost by themselves)
            else if (TRUNC == 8) s_base = tot ? atomicAdd(&cnt[1].trunc_pad, tot) : 0ull;   // (8 = 4 with ONE head for everybody, as the kernel was before)
            else
            // (a truncated instantiation reserves from the head's padding: the same atomic traffic, nothing the real kernel reads)
            s_base = tot ? atomicAdd(reinterpret_cast<unsigned long long*>(TRUNC ? &head->pad[0] : &head->n_paths), (tot & 0xffffffffull) | ((unsigned long long)(np * n_nee) << 32)) : 0ull;
        }
        __syncthreads();
        unsigned long long base = s_base;
        const uint32_t stride = s_stride;
        for (uint32_t k = 0; k < wave; k++) base += s_cnt[k];
        __syncthreads();   // s_cnt / s_base / s_stride are rewritten by the next iteration
        const uint32_t kq = (base_i >> QUEUE_TILE_SHIFT) & (QUEUE_SUBS - 1u);
        const uint32_t j = queue_slot((uint32_t)base + (uint32_t)__popcll(ma & lt), kq);
        const uint32_t q = (uint32_t)(base >> 32) + (uint32_t)__popcll(mn & lt);   // sample k of this path: entry q + k * stride of sub-queue kq = queue slot queue_slot(q + k * stride, kq)
        if (TRUNC == 4 || TRUNC == 7 || TRUNC == 8) { acc ^= j ^ q ^ stride; continue; }
        // (TRUNC 5 / 6: the stores below become folds into acc)
        auto st4 = [&](float4* p_, float4 v_) { if (TRUNC == 0) nt_store(p_, v_); else acc ^= f2u(v_.x) ^ f2u(v_.y) ^ f2u(v_.z) ^ f2u(v_.w) ^ (uint32_t)(uintptr_t)p_; };
        auto st2 = [&](uint2* p_, uint2 v_) { if (TRUNC == 0) nt_store(p_, v_); else acc ^= v_.x ^ v_.y ^ (uint32_t)(uintptr_t)p_; };
        // ---- phase C: light samples (integrator.hlsl:137-151) and the next direction (:153-165) ----
        if (SPEC != 1 && alive) {
            uint32_t valid = 0;
            if (SPEC != 2 && !delta) {   // the random numbers of every sample are drawn whether or not it gets a queue entry
                for (uint32_t k = 0; k < env_n; k++) {   // integrator.hlsl:141-144
                    f2 rand; rand.x = rng_float(rng); rand.y = rng_float(rng);
                    const LSample ls = env_sample_unoccluded(sc.env, rand, MSNE_ENV_TOP_LDS ? s_envtop : nullptr);
                    const uint32_t e_ = queue_slot(q + k * stride, kq);
                    if (ls.pdf > 0.0f) {
                        const f3 so = offset_along_normal(attrs.position, face_forward(attrs.triangleFrame.n, ls.dirWs));
                        const f3 e = estimate_direct_mis(shadingFrame, ls, material, woSs, env_n);
                        const f3 cc = divs(mul(throughput, e), (float)env_n);
                        st4(&shq.o[e_], make_float4(so.x, so.y, so.z, INFINITY_F)); st4(&shq.d[e_], make_float4(ls.dirWs.x, ls.dirWs.y, ls.dirWs.z, 0.0f));
                        st4(&shq.c[e_], make_float4(cc.x, cc.y, cc.z, 0.0f));
                        valid++;
                    } else { st4(&shq.o[e_], make_float4(0.0f, 0.0f, 0.0f, -1.0f)); st4(&shq.c[e_], make_float4(0.0f, 0.0f, 0.0f, 0.0f)); }
                }
                for (uint32_t k = 0; k < mesh_n; k++) {  // integrator.hlsl:147-150 + MeshLights::sample light.hlsl:130-158
                    f2 rand; rand.x = rng_float(rng); rand.y = rng_float(rng);
                    const uint32_t e_ = queue_slot(q + (env_n + k) * stride, kq);
                    if (!have_lights) continue;
                    bool ok = false;
                    const uint32_t entryCount = sc.alias_count;
                    const float sum = sc.alias_sum;
                    {
                        const float scaled = rand.x * (float)entryCount;
                        uint32_t idx = (uint32_t)scaled;
                        rand.x = scaled - floor_(scaled);
                        const AliasEntry en = alias_load(sc, entryCount, 1 + idx);
                        if (!coin_flip_remap(en.select, rand.x)) idx = en.alias;
                        const f2 bary = square_to_triangle(rand);
                        // MeshAttributes::lookupAndInterpolate(...).inWorld(...) (world.hlsl:114-176) from the light's gathered record
                        // (position, texcoord and triangle normal are all the light sample uses; same operations, same order)
                        const uint4* lp = reinterpret_cast<const uint4*>(sc.light_tris + (idx < entryCount ? idx : entryCount));
                        const uint4 la = lp[0], lb = lp[1], lc = lp[2], ld = lp[3], le = lp[4], lf = lp[5], lg = lp[6], lh = lp[7], li = lp[8], lj = lp[9];
                        m34 ltoWorld;
                        ltoWorld.m[0][0] = u2f(lf.x); ltoWorld.m[0][1] = u2f(lf.y); ltoWorld.m[0][2] = u2f(lf.z); ltoWorld.m[0][3] = u2f(lf.w);
                        ltoWorld.m[1][0] = u2f(lg.x); ltoWorld.m[1][1] = u2f(lg.y); ltoWorld.m[1][2] = u2f(lg.z); ltoWorld.m[1][3] = u2f(lg.w);
                        ltoWorld.m[2][0] = u2f(lh.x); ltoWorld.m[2][1] = u2f(lh.y); ltoWorld.m[2][2] = u2f(lh.z); ltoWorld.m[2][3] = u2f(lh.w);
                        TexDesc t_light; t_light.offset = li.x; t_light.w = li.y; t_light.h = li.z; t_light.format = li.w; t_light.first = make_float4(u2f(lj.x), u2f(lj.y), u2f(lj.z), u2f(lj.w));
                        const f3 lp0 = F3(u2f(la.x), u2f(la.y), u2f(la.z)), lp1 = F3(u2f(la.w), u2f(lb.x), u2f(lb.y)), lp2 = F3(u2f(lb.z), u2f(lb.w), u2f(lc.x));
                        const f3 lbary = F3(1.0f - bary.x - bary.y, bary.x, bary.y);
                        const f3 at_position = m34_mul_point(ltoWorld, interp3(lbary, lp0, lp1, lp2));
                        const f2 at_texcoord = interp2(lbary, F2(u2f(lc.y), u2f(lc.z)), F2(u2f(lc.w), u2f(ld.x)), F2(u2f(ld.y), u2f(ld.z)));
                        const f3 at_n = F3(u2f(le.x), u2f(le.y), u2f(le.z));
                        LSample ls;
                        { const float4 em = tex_sample_desc<TEX>(sc, t_light, at_texcoord); ls.radiance = F3(em.x, em.y, em.z); }
                        ls.dirWs = normalize(sub(at_position, attrs.position));
                        ls.pdf = area_to_solid_angle(at_position, attrs.position, ls.dirWs, at_n) / sum;
                        if (ls.pdf > 0.0f) {
                            const f3 offL = offset_along_normal(at_position, at_n);
                            const f3 offS = offset_along_normal(attrs.position, face_forward(attrs.triangleFrame.n, ls.dirWs));
                            const float st_ = length(sub(offL, offS)); const f3 sd = normalize(sub(offL, offS));
                            const f3 e = estimate_direct_mis(shadingFrame, ls, material, woSs, mesh_n);
                            const f3 cc = divs(mul(throughput, e), (float)mesh_n);
                            st4(&shq.o[e_], make_float4(offS.x, offS.y, offS.z, st_)); st4(&shq.d[e_], make_float4(sd.x, sd.y, sd.z, 0.0f));
                            st4(&shq.c[e_], make_float4(cc.x, cc.y, cc.z, 0.0f));
                            ok = true; valid++;
                        }
                    }
                    if (!ok) { st4(&shq.o[e_], make_float4(0.0f, 0.0f, 0.0f, -1.0f)); st4(&shq.c[e_], make_float4(0.0f, 0.0f, 0.0f, 0.0f)); }
                }
            }
            if (TRUNC == 5) { acc ^= valid ^ rng; continue; }
            // next direction, integrator.hlsl:153-165
            f2 sq; sq.x = rng_float(rng); sq.y = rng_float(rng);
            const MSample sample = material_sample(material, woSs, sq);
            if (sample.pdf == 0.0f) {
                // the path ends here; with light samples in flight it is finalised one pass later (after their shadow rays)
                if (TRUNC == 0) atomicAdd(&cnt[1].zombies, 1u);
                if (valid) { st4(&nxt.ro[j], make_float4(0.0f, 0.0f, 0.0f, u2f(PATH_FLAG_ZOMBIE | PATH_FLAG_NEE | (stride << PATH_STRIDE_SHIFT)))); st4(&nxt.lr[j], make_float4(L.x, L.y, L.z, 0.0f)); st2(&nxt.sq[j], make_uint2(slot, shq_pack(q, kq))); }
                else { st4(&nxt.ro[j], make_float4(0.0f, 0.0f, 0.0f, u2f(PATH_FLAG_ZOMBIE | PATH_FLAG_DEAD))); st4(&lbuf[slot], make_float4(L.x, L.y, L.z, 0.0f)); }
            } else {
                const f3 nd = frame_frame_to_world(shadingFrame, sample.dirFs);
                const f3 no = offset_along_normal(attrs.position, face_forward(attrs.triangleFrame.n, nd));
                const f3 f = material_eval(material, sample.dirFs, woSs);
                const float ac = absf(sample.dirFs.z);
                const f3 tp = mul(throughput, F3(f.x * ac / sample.pdf, f.y * ac / sample.pdf, f.z * ac / sample.pdf));
                const uint32_t nf = ((bounceCount + 1u) & 0xFFFFu) | (delta ? PATH_FLAG_DELTA : 0u) | (nee ? (PATH_FLAG_NEE | (stride << PATH_STRIDE_SHIFT)) : 0u);
                st4(&nxt.ro[j], make_float4(no.x, no.y, no.z, u2f(nf))); st4(&nxt.rd[j], make_float4(nd.x, nd.y, nd.z, 0.0f));
                st4(&nxt.tp[j], make_float4(tp.x, tp.y, tp.z, sample.pdf)); st4(&nxt.lr[j], make_float4(L.x, L.y, L.z, u2f(rng))); st2(&nxt.sq[j], make_uint2(slot, shq_pack(q, kq)));
            }
        }
    }
    if (TRUNC != 0 && acc == 0x9e3779b9u && cnt[0].head_closest == 0xffffffffu) lbuf[0] = make_float4(u2f(acc), 0.0f, 0.0f, 0.0f);   // (never — but only memory knows: keeps the truncated kernel's loads alive)
}

// statistics of a finished batch: rays traced and camera paths started, from its per-bounce counters
__global__ void k_account(const BounceCounters* c, uint32_t n_bounces, Totals* t) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long closest = 0, shadow = 0;
    for (uint32_t b = 0; b <= n_bounces; b++) { closest += queue_total(&c[b], false) - c[b].zombies; shadow += c[b].n_shadow_traced; }
    t->closest_rays += closest; t->shadow_rays += shadow; t->samples += queue_total(&c[0], false) - c[0].zombies;
}

// storeColor main.hlsl:43-51 over the `s_count` samples of this chunk (summed in sample order, main.hlsl:83-92)
__global__ __launch_bounds__(SHADE_BLOCK) void k_film(ShardView sh, PipelineOpts opts, const float4* lbuf, uint32_t s_count, uint32_t n_launches, int first_chunk, int last_chunk,
                                                        uint32_t sample_count, float4* color, float4* film) {
    // lbuf holds n_launches launches of s_count samples each (slot = (launch*s_count + sample)*pixels + p).  The launches were
    // traced concurrently but are folded into the film here in launch order, exactly as n_launches sequential dispatches
    // would (main.hlsl:43-51).  A launch that does not fit in flight is split into chunks (then n_launches == 1 and
    // first_chunk/last_chunk tell which part of its sample sum this is).
    for (uint32_t p = blockIdx.x * SHADE_BLOCK + threadIdx.x; p < sh.pixels; p += gridDim.x * SHADE_BLOCK) {
        uint32_t x, y;
        if (!shard_pixel(sh, p, x, y)) continue;
        float4 f = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (last_chunk && sample_count != 0) f = film[p];
        for (uint32_t j = 0; j < n_launches; j++) {
            f3 c = F3(0.0f, 0.0f, 0.0f);
            if (!first_chunk) { const float4 q = color[p]; c = F3(q.x, q.y, q.z); }
            for (uint32_t s = 0; s < s_count; s++) { const float4 l = nt_load(&lbuf[((size_t)j * s_count + s) * sh.pixels + p]); c = add(c, F3(l.x, l.y, l.z)); }
            if (!last_chunk) { color[p] = make_float4(c.x, c.y, c.z, 0.0f); continue; }
            const float spr = (float)opts.samples_per_run;
            const uint32_t count = sample_count + j * opts.samples_per_run;
            if (count == 0) f = make_float4(c.x / spr, c.y / spr, c.z / spr, 1.0f);
            else {
                const float den = (float)(count + opts.samples_per_run);
                f = make_float4(f.x + (c.x - f.x) / den, f.y + (c.y - f.y) / den, f.z + (c.z - f.z) / den, f.w + 1.0f);
            }
        }
        if (last_chunk) film[p] = f;
    }
}

// packed (tile-major, per shard) film → row-major full film.  `packed` holds `nshards` shard films back to back,
// each `stride` float4 long (MsneUnpackGatheredFilm); nshards == 1 for the local readback.
__global__ __launch_bounds__(SHADE_BLOCK) void k_unpack_film(ShardView sh, const float4* packed, uint32_t nshards, uint32_t first_shard, size_t stride, float4* full) {
    const uint32_t tpix = sh.tile_size * sh.tile_size;
    const uint32_t total_tiles = sh.tiles_x * sh.tiles_y;
    for (size_t g = (size_t)blockIdx.x * SHADE_BLOCK + threadIdx.x; g < (size_t)total_tiles * tpix; g += (size_t)gridDim.x * SHADE_BLOCK) {
        const uint32_t t = (uint32_t)(g / tpix), w = (uint32_t)(g % tpix);
        const uint32_t shard = t % sh.shard_count, k = t / sh.shard_count;
        if (shard < first_shard || shard >= first_shard + nshards) continue;
        const uint32_t x = (t % sh.tiles_x) * sh.tile_size + w % sh.tile_size, y = (t / sh.tiles_x) * sh.tile_size + w / sh.tile_size;
        if (x >= sh.width || y >= sh.height) continue;
        full[(size_t)y * sh.width + x] = packed[(size_t)(shard - first_shard) * stride + (size_t)k * tpix + w];
    }
}

// Batch probe of the shading functions for parity tests (MsneShadeProbe): the SAME device functions k_shade runs, one record
// per thread.  The table (function codes, record widths) is the test oracle's OrcProbeBatch.
__constant__ uint32_t c_probe_in[21]  = { 15, 2, 3, 3, 2, 3, 2, 2, 2, 3, 6, 3, 12, 7, 7, 4, 9, 3, 51, 13, 23 };
__constant__ uint32_t c_probe_out[21] = {  8, 7, 4, 3, 3, 2, 2, 2, 3, 1, 3, 6,  1, 3, 3, 1, 6, 4, 23,  9,  6 };
__global__ void k_shade_probe(SceneView sc, int fn, const float* in, uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = in + (size_t)i * c_probe_in[fn]; float* o = out + (size_t)i * c_probe_out[fn];
    switch (fn) {
        case 0: {
            Mat m; m.type = (uint32_t)a[0]; m.color = F3(a[1], a[2], a[3]); m.metalness = a[4]; m.alpha = maxf(a[5] * a[5], 0.001f); m.ior = a[6];
            const f3 wi = F3(a[7], a[8], a[9]), wo = F3(a[10], a[11], a[12]);
            o[0] = material_pdf(m, wi, wo);
            const f3 e = material_eval(m, wi, wo); o[1] = e.x; o[2] = e.y; o[3] = e.z;
            const MSample s = material_sample(m, wo, F2(a[13], a[14])); o[4] = s.dirFs.x; o[5] = s.dirFs.y; o[6] = s.dirFs.z; o[7] = s.pdf;
            break; }
        case 1: { const LSample s = env_sample_unoccluded(sc.env, F2(a[0], a[1])); o[0] = s.dirWs.x; o[1] = s.dirWs.y; o[2] = s.dirWs.z; o[3] = s.radiance.x; o[4] = s.radiance.y; o[5] = s.radiance.z; o[6] = s.pdf; break; }
        case 2: { f3 r; float p; env_eval(sc.env, F3(a[0], a[1], a[2]), r, p); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = p; break; }
        case 3: { const f3 r = env_incoming_radiance(sc.env, F3(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
        case 4: { const f3 d = square_to_equal_area_sphere(F2(a[0], a[1])); o[0] = d.x; o[1] = d.y; o[2] = d.z; break; }
        case 5: { const f2 u = square_to_equal_area_sphere_inverse(F3(a[0], a[1], a[2])); o[0] = u.x; o[1] = u.y; break; }
        case 6: { const f2 u = square_to_triangle(F2(a[0], a[1])); o[0] = u.x; o[1] = u.y; break; }
        case 7: { const f2 u = square_to_gaussian(F2(a[0], a[1])); o[0] = u.x; o[1] = u.y; break; }
        case 8: { const f3 d = square_to_cosine_hemisphere(F2(a[0], a[1])); o[0] = d.x; o[1] = d.y; o[2] = d.z; break; }
        case 9: o[0] = fresnel_dielectric(a[0], a[1], a[2]); break;
        case 10: { const f3 r = offset_along_normal(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5])); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
        case 11: { f3 p, q; coordinate_system(F3(a[0], a[1], a[2]), p, q); o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; break; }
        case 12: o[0] = area_to_solid_angle(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), F3(a[6], a[7], a[8]), F3(a[9], a[10], a[11])); break;
        case 13: o[0] = ggx_D(a[0], F3(a[1], a[2], a[3])); o[1] = ggx_Lambda(a[0], F3(a[1], a[2], a[3])); o[2] = ggx_G(a[0], F3(a[1], a[2], a[3]), F3(a[4], a[5], a[6])); break;
        case 14: { const f3 r = refract_dir(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), a[6]); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
        case 15: o[0] = power_heuristic((uint32_t)a[0], a[1], (uint32_t)a[2], a[3]); break;
        case 16: { Frame f; f.n = F3(a[0], a[1], a[2]); f.s = F3(a[3], a[4], a[5]); f.t = F3(0.0f, 0.0f, 0.0f); frame_reorthogonalize(f);
                   const f3 p = frame_world_to_frame(f, F3(a[6], a[7], a[8])), q = frame_frame_to_world(f, F3(a[6], a[7], a[8]));
                   o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; break; }
        case 17: { const float4 t = tex_sample(sc, (uint32_t)a[0], F2(a[1], a[2])); o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w; break; }   // (the caller checked the index)
        case 18: {   // MeshAttributes::lookupAndInterpolate(...).inWorld(...) on explicit vertex data: flags bit 0 = the mesh has texcoords, bit 1 = normals
            const uint32_t flags = (uint32_t)a[26];
            const bool ht = (flags & 1u) != 0u, hn = (flags & 2u) != 0u;
            const f2 t0 = ht ? F2(a[9], a[10]) : F2(0.0f, 0.0f), t1 = ht ? F2(a[11], a[12]) : F2(1.0f, 0.0f), t2 = ht ? F2(a[13], a[14]) : F2(1.0f, 1.0f);
            m34 tw, tm;
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) { tw.m[r][c] = a[27 + 4 * r + c]; tm.m[r][c] = a[39 + 4 * r + c]; }
            const Attrs at = mesh_attributes_core(F3(a[0], a[1], a[2]), F3(a[3], a[4], a[5]), F3(a[6], a[7], a[8]), t0, t1, t2, F3(a[15], a[16], a[17]), F3(a[18], a[19], a[20]), F3(a[21], a[22], a[23]),
                                                  hn, F3(1.0f - a[24] - a[25], a[24], a[25]), tw, tm);
            o[0] = at.position.x; o[1] = at.position.y; o[2] = at.position.z; o[3] = at.texcoord.x; o[4] = at.texcoord.y;
            const Frame* fr[2] = { &at.triangleFrame, &at.frame };
            for (int k = 0; k < 2; k++) { float* q = o + 5 + 9 * k; q[0] = fr[k]->n.x; q[1] = fr[k]->n.y; q[2] = fr[k]->n.z; q[3] = fr[k]->s.x; q[4] = fr[k]->s.y; q[5] = fr[k]->s.z; q[6] = fr[k]->t.x; q[7] = fr[k]->t.y; q[8] = fr[k]->t.z; }
            break; }
        case 19: {   // getTextureFrame on an already sampled normal texel
            Frame tf; tf.n = F3(a[3], a[4], a[5]); tf.s = F3(a[6], a[7], a[8]); tf.t = F3(a[9], a[10], a[11]);
            const Frame f = texture_frame_from_texel(make_float4(a[0], a[1], a[2], 1.0f), a[12] != 0.0f, tf);
            o[0] = f.n.x; o[1] = f.n.y; o[2] = f.n.z; o[3] = f.s.x; o[4] = f.s.y; o[5] = f.s.z; o[6] = f.t.x; o[7] = f.t.y; o[8] = f.t.z;
            break; }
        case 20: {   // Camera::generateRay: the host evaluated make_camera() on the lens (as MsneRender does); 19 constants, uv, rand
            CameraConsts cam;
            cam.origin = F3(a[0], a[1], a[2]); cam.u = F3(a[3], a[4], a[5]); cam.v = F3(a[6], a[7], a[8]); cam.horizontal = F3(a[9], a[10], a[11]);
            cam.vertical = F3(a[12], a[13], a[14]); cam.llc = F3(a[15], a[16], a[17]); cam.aperture = a[18];
            f3 O, D; camera_generate_ray(cam, F2(a[19], a[20]), F2(a[21], a[22]), O, D);
            o[0] = O.x; o[1] = O.y; o[2] = O.z; o[3] = D.x; o[4] = D.y; o[5] = D.z;
            break; }
    }
}
bool shade_probe_widths(int fn, uint32_t& win, uint32_t& wout) {
    // (fn 20: the width on the device — the host turns the caller's 18-float lens record into make_camera()'s constants first)
    static const uint32_t pin[21]  = { 15, 2, 3, 3, 2, 3, 2, 2, 2, 3, 6, 3, 12, 7, 7, 4, 9, 3, 51, 13, 23 }, pout[21] = { 8, 7, 4, 3, 3, 2, 2, 2, 3, 1, 3, 6, 1, 3, 3, 1, 6, 4, 23, 9, 6 };
    if (fn < 0 || fn > 20) return false;
    win = pin[fn]; wout = pout[fn]; return true;
}
void launch_shade_probe(hipStream_t s, const SceneView& sc, int fn, const float* in, uint32_t n, float* out) {
    hipLaunchKernelGGL(k_shade_probe, dim3((n + 63) / 64), dim3(64), 0, s, sc, fn, in, n, out);
}

// ---------------- host launch wrappers ----------------
void launch_raygen(hipStream_t s, int grid, const ShardView& sh, const CameraConsts& cam, const PipelineOpts& o, uint32_t sample_base, uint32_t s_count, const PathState& st, BounceCounters* cnt) {
    hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(SHADE_BLOCK), 0, s, sh, cam, o, sample_base, s_count, st, cnt);
}
void launch_shade(hipStream_t s, int grid, const SceneView& sc, const PipelineOpts& o, const PathState& cur, const HitBuf& hits, const PathState& nxt, const ShadowQueue& q, const float4* c_prev, float4* lbuf, BounceCounters* cnt, bool first_pass, bool textured) {
    static const bool specialised = [] { const char* e = getenv("MSNE_SHADE_SPEC"); return e && atoi(e) != 0; }();
    const uint32_t fp = first_pass ? 1u : 0u;
    static const bool trunc = [] { const char* e = getenv("MSNE_SHADE_TRUNC"); return e && atoi(e) != 0; }();
    if (trunc && !first_pass) {   // measurement: the truncated kernels on the same queues, in front of the real one
#define MSNE_TRUNC_LAUNCH(T) do { if (textured) hipLaunchKernelGGL((k_shade<0, true, T>), dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o, cur, hits, nxt, q, c_prev, lbuf, cnt, fp); \
                                  else hipLaunchKernelGGL((k_shade<0, false, T>), dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o, cur, hits, nxt, q, c_prev, lbuf, cnt, fp); } while (0)
        MSNE_TRUNC_LAUNCH(1); MSNE_TRUNC_LAUNCH(2); MSNE_TRUNC_LAUNCH(3); MSNE_TRUNC_LAUNCH(4); MSNE_TRUNC_LAUNCH(7); MSNE_TRUNC_LAUNCH(8); MSNE_TRUNC_LAUNCH(5); MSNE_TRUNC_LAUNCH(6);
#undef MSNE_TRUNC_LAUNCH
    }
    if (!specialised) {
        if (textured) hipLaunchKernelGGL((k_shade<0, true>), dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o, cur, hits, nxt, q, c_prev, lbuf, cnt, fp);
        else hipLaunchKernelGGL((k_shade<0, false>), dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o, cur, hits, nxt, q, c_prev, lbuf, cnt, fp);
        return;
    }
    hipLaunchKernelGGL((k_shade<1, true>), dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o, cur, hits, nxt, q, c_prev, lbuf, cnt, fp);
    hipLaunchKernelGGL((k_shade<2, true>), dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o, cur, hits, nxt, q, c_prev, lbuf, cnt, fp);
    hipLaunchKernelGGL((k_shade<3, true>), dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o, cur, hits, nxt, q, c_prev, lbuf, cnt, fp);
    hipLaunchKernelGGL((k_shade<4, true>), dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o, cur, hits, nxt, q, c_prev, lbuf, cnt, fp);
}
void launch_light_tris(hipStream_t s, const SceneView& sc, uint32_t indexed_attributes, uint32_t n_instances, LightTri* out) {
    hipLaunchKernelGGL(k_light_tris, dim3((sc.alias_count + 1 + 255) / 256), dim3(256), 0, s, sc, indexed_attributes, n_instances, out);
}
void launch_account(hipStream_t s, const BounceCounters* c, uint32_t n_bounces, Totals* t) { hipLaunchKernelGGL(k_account, dim3(1), dim3(1), 0, s, c, n_bounces, t); }
void launch_film(hipStream_t s, int grid, const ShardView& sh, const PipelineOpts& o, const float4* lbuf, uint32_t s_count, uint32_t n_launches, int first_chunk, int last_chunk, uint32_t sample_count, float4* color, float4* film) {
    hipLaunchKernelGGL(k_film, dim3(grid), dim3(SHADE_BLOCK), 0, s, sh, o, lbuf, s_count, n_launches, first_chunk, last_chunk, sample_count, color, film);
}
void launch_unpack_film(hipStream_t s, int grid, const ShardView& sh, const float4* packed, uint32_t nshards, uint32_t first_shard, size_t stride, float4* full) {
    hipLaunchKernelGGL(k_unpack_film, dim3(grid), dim3(SHADE_BLOCK), 0, s, sh, packed, nshards, first_shard, stride, full);
}

}  // namespace msne
