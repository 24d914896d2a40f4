// msne_device.h — HBM data layout of the hot path.
#pragma once
#include "msne_math.h"

#include <cstdlib>

namespace msne {

// $MSNE_DEBUG_POISON (anything but "0"): every device allocation and the builder's scratch start as 0xCD garbage, as recycled device memory does in a long-lived
// process — a value read before it is written shows.  The tests run that way (tests/conftest.py) and run a subset without it ("0"): fresh memory, no extra syncs.
inline bool debug_poison() { const char* e = getenv("MSNE_DEBUG_POISON"); return e && !(e[0] == '0' && e[1] == 0); }

// ---- acceleration structure ----
// 8-wide quantized node, 80 B = 5 x 16 B.  Child boxes: lo = origin + qlo * 2^(e-127) per axis.
// Children sit in octant order (slot bit k set = the child lies on the + side of axis k of the node).
// imask bit i: slot i is an internal node at nodes[child_base + popcount(imask & ((1<<i)-1))];
// lmask bit i: slot i is a leaf holding ONE item (triangle record / TLAS instance) at item_base + popcount(lmask & ((1<<i)-1));
// neither: empty slot (inverted box).
struct alignas(16) Node8 {
    float ox, oy, oz;
    uint8_t ex, ey, ez, imask;
    uint32_t child_base;
    uint32_t item_base;
    uint8_t lmask, pad[7];
    uint8_t qlo[3][8];
    uint8_t qhi[3][8];
};
static_assert(sizeof(Node8) == 80, "Node8 must be 80 bytes");

// triangle record in BVH order, object space (48 B = 3 x 16 B)
struct alignas(16) TriRec {
    float v0x, v0y, v0z, v1x;
    float v1y, v1z, v2x, v2y;
    float v2z; uint32_t geo, prim, pad;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 bytes");

// The same three vertices once more, for the traversal kernels only (same slot of a parallel pool): every vertex as x y z x y, so that the three dwords at offset r
// are the vertex's coordinates rotated by r — (v[r], v[r + 1], v[r + 2]) for r = the ray's dominant axis kz.  The watertight test permutes every vertex by the ray's
// dominant axis; with this record the permutation is the ADDRESS of three 12-B loads instead of 18 v_cndmask + 6 v_cmp per test (a quarter of the test's issue
// cycles).  `inst` repeats TriRec::pad.  64 B, never straddles a 128-B line.  (Measured against the alternative without a second pool — the 48-B record stored by axis,
// {x0 x1 x2}{y0 y1 y2}{z0 z1 z2}, its three blocks loaded in rotated order: three 64-bit address computations per test instead of one cost more than the selects had
// on S1 (-1.1 % against +1.2 % for this record); profiles/r05_tri_density.txt.)
struct alignas(16) TriRot { float v[15]; uint32_t inst; };
static_assert(sizeof(TriRot) == 64, "TriRot must be 64 bytes");

// What MeshAttributes::lookupAndInterpolate (world.hlsl:114-158) reads beyond the positions, for the triangle in the SAME slot of the
// triangle pool: the three normals and texcoords the pipeline's attribute mode selects (by vertex index for glTF, by corner for Hydra),
// gathered once at build time.  A hit then needs one 64-B record instead of the chain geometry -> mesh -> indices -> 6 scattered attributes.
// Only filled (and only read) for geometries whose mesh has normals or texcoords.
struct alignas(16) TriAttr { float n0x, n0y, n0z, n1x, n1y, n1z, n2x, n2y, n2z, t0x, t0y, t1x, t1y, t2x, t2y, pad; };
static_assert(sizeof(TriAttr) == 64, "TriAttr must be 64 bytes");

// per instance (Accel.zig:394-432): object->world, world->object, first geometry, BLAS root
struct alignas(16) InstanceRec {
    m34 transform;
    m34 world_to_instance;
    uint32_t geo_offset;   // instanceID() (world.hlsl:12-14)
    uint32_t blas_root;    // node index of the BLAS root, MAX_UINT if the BLAS is empty
    uint32_t flags;        // bit0 visible, bit1 identity transform
    uint32_t pad;
};
static_assert(sizeof(InstanceRec) == 112, "InstanceRec size");

// world.hlsl:19-23 (12 B).  `sampled`: bit 0 = sampled for emitted light (the reference's bool32), bits 1/2 = the mesh has
// texcoords / normals (lets k_shade skip the MeshRec load for bare meshes)
struct GeometryRec { uint32_t mesh, material, sampled; };
constexpr uint32_t GEO_SAMPLED = 1u, GEO_HAS_TEXCOORDS = 2u, GEO_HAS_NORMALS = 4u;
struct MeshRec { const float* positions; const float* texcoords; const float* normals; const uint32_t* indices; }; // world.hlsl:25-31 (32 B)
// reference: Material{normal,emissive,type,addr}+variant buffer (MaterialManager.zig:35-77); flattened here to one 32-B record
struct alignas(16) MaterialRec { uint32_t normal, emissive, type, color, metalness, roughness; float ior; uint32_t pad; };
// Textures stay in HBM in the format they were created with, as the reference uploads them in their own vk.Format (MaterialManager.zig:351-390):
// `format` is a MsneTextureFormat, `offset` counts 16-B units into SceneView::texels, a lookup decodes the texels it touches (texel_decode, shade.h).
// `first` repeats texel 0 DECODED: 1x1 textures — every constant material parameter (World.zig:44-228) — are served by the descriptor load alone.
constexpr uint32_t TEX_RGBA8_SRGB = 0, TEX_RG8_UNORM = 1, TEX_R8_UNORM = 2, TEX_RGBA32F = 3, TEX_RG32F = 4, TEX_R32F = 5, TEX_RGBA16F = 6;   // == MsneTextureFormat
MSNE_HD uint32_t tex_bytes_per_texel(uint32_t format) { return format == TEX_RGBA8_SRGB ? 4u : format == TEX_RG8_UNORM ? 2u : format == TEX_R8_UNORM ? 1u : format == TEX_RGBA32F ? 16u : format == TEX_RG32F ? 8u : format == TEX_R32F ? 4u : 8u; }
struct alignas(16) TexDesc { uint32_t offset, w, h, format; float4 first; };
MSNE_HD float half_bits_to_float(uint32_t h) {   // IEEE binary16 -> binary32, exact (subnormals included)
    const uint32_t s = (h >> 15) << 31, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0u) { if (m == 0u) return u2f(s); const float f = (float)m * 0x1p-24f; return (h >> 15) ? -f : f; }
    if (e == 31u) return u2f(s | 0x7f800000u | (m << 13));
    return u2f(s | ((e + 112u) << 23) | (m << 13));
}
// One texel as the float RGBA the reference's sampler would return for the format: sRGB bytes through the 256-entry table `srgb` the host computed,
// UNORM bytes / 255, halves and floats as they are; missing channels 0, 0, 0, 1.  Host and device evaluate the same expressions: the descriptor's
// `first` texel, the test oracle's upload-time decode and a lookup on the GPU agree bit for bit.
MSNE_HD float4 decode_rgba8_srgb(uint32_t p, const float* srgb) { float4 o; o.x = srgb[p & 255u]; o.y = srgb[(p >> 8) & 255u]; o.z = srgb[(p >> 16) & 255u]; o.w = (float)(p >> 24) / 255.0f; return o; }
MSNE_HD float4 decode_rg8_unorm(uint32_t p) { float4 o; o.x = (float)(p & 255u) / 255.0f; o.y = (float)((p >> 8) & 255u) / 255.0f; o.z = 0.0f; o.w = 1.0f; return o; }
MSNE_HD float4 decode_r8_unorm(uint32_t p) { float4 o; o.x = (float)(p & 255u) / 255.0f; o.y = 0.0f; o.z = 0.0f; o.w = 1.0f; return o; }
MSNE_HD float4 decode_rgba16f(uint32_t lo, uint32_t hi) { float4 o; o.x = half_bits_to_float(lo & 0xffffu); o.y = half_bits_to_float(lo >> 16); o.z = half_bits_to_float(hi & 0xffffu); o.w = half_bits_to_float(hi >> 16); return o; }
// texel `idx` of a texture whose first byte is `base`
MSNE_HD float4 texel_decode(const uint8_t* base, uint32_t format, size_t idx, const float* srgb) {
    float4 o; o.x = 0.0f; o.y = 0.0f; o.z = 0.0f; o.w = 1.0f;
    switch (format) {
        case TEX_RGBA8_SRGB: return decode_rgba8_srgb(reinterpret_cast<const uint32_t*>(base)[idx], srgb);
        case TEX_RG8_UNORM: return decode_rg8_unorm(reinterpret_cast<const uint16_t*>(base)[idx]);
        case TEX_R8_UNORM: return decode_r8_unorm(base[idx]);
        case TEX_RGBA32F: { const float* f = reinterpret_cast<const float*>(base) + 4 * idx; o.x = f[0]; o.y = f[1]; o.z = f[2]; o.w = f[3]; break; }
        case TEX_RG32F: { const float* f = reinterpret_cast<const float*>(base) + 2 * idx; o.x = f[0]; o.y = f[1]; break; }
        case TEX_R32F: o.x = reinterpret_cast<const float*>(base)[idx]; break;
        default: { const uint32_t* h = reinterpret_cast<const uint32_t*>(base) + 2 * idx; return decode_rgba16f(h[0], h[1]); }
    }
    return o;
}
struct AliasEntry { uint32_t alias; float select; uint32_t instance, geometry, primitive; };  // light.hlsl:17-22,112-116 (20 B)
// What MeshLights::sample (light.hlsl:130-158) needs of alias entry i, gathered once per scene: the object-space vertices and
// texcoords of the emissive triangle and its material — one record per light sample instead of the chain
// instance → geometry → mesh → indices → positions.  Entry [count] stands for the all-zero entry of an out-of-range index.
struct alignas(16) LightTri {
    float p0x, p0y, p0z, p1x, p1y, p1z, p2x, p2y, p2z, t0x, t0y, t1x, t1y, t2x, t2y; uint32_t material;
    // ... and what else MeshLights::sample (light.hlsl:130-158) fetches through indices after it: the triangle's world-space normal exactly as
    // inWorld() derives it (it does not depend on the sampled point), the instance's object-to-world matrix, and the descriptor of the material's
    // emissive texture.  A light sample is then alias entry -> this record -> (texel), not alias -> alias -> record -> instance -> material -> descriptor -> texel.
    float nx, ny, nz; uint32_t instance;
    float to_world[12];
    TexDesc emissive;
};
static_assert(sizeof(LightTri) == 160, "LightTri must be 160 bytes");

struct EnvView {
    const float4* rgb;        // S*S equal-area map
    const float* lum;         // luminance pyramid, level l at lum + lum_offset[l], (S>>l)^2 texels
    uint32_t lum_offset[12];
    uint32_t size, mip_count;
    // The same pyramid once more, packed for EnvMap::sample's descent: level l (2x2 and larger) as one float4 per 2x2 quad, {(x,y), (x,y+1), (x+1,y), (x+1,y+1)}
    // of the quad whose corner is (x, y) = 2 * (parent texel) — everything one level of the descent reads, in one 16-B load instead of two dependent rounds of loads.
    const float4* quads;      // level l at quads + quad_offset[l], ((S>>l)/2)^2 quads; nullptr for a 1x1 map
    uint32_t quad_offset[12];
    float top;                // the single texel of the last level (the map's integral)
};

// One record per TLAS leaf: everything a ray needs to enter the instance behind it (Accel.zig:394-432's instance record cut to the traversal's part) in ONE 64-B
// fetch instead of tlas_items -> InstanceRec.  root = MAX_UINT: nothing to enter (hidden, empty BLAS); inst = WORLD_INSTANCE for the merged world BLAS.
// + the instance's WORLD-SPACE BOUNDING SPHERE (centre of its box, the farthest transformed vertex from it, a little slack): a ray that passes the leaf's box test is tested
// against the sphere before it pays for the change of space and the visit of the BLAS root — on S2 (spheres in boxes) that turns away four entries in ten (round 5).
struct alignas(16) TlasLeaf { float w2i[12]; uint32_t root, inst, flags, pad; float sx, sy, sz, sr; };
static_assert(sizeof(TlasLeaf) == 80, "TlasLeaf must be 80 bytes");

struct SceneView {
    const Node8* nodes;
    const TriRec* tris;
    const TriRot* tri_rot;            // parallel to tris (same slot): what the traversal kernels read
    const TriAttr* tri_attrs;         // parallel to tris (same slot); nullptr when no mesh of the scene has normals or texcoords
    const uint32_t* tlas_items;       // instance index per TLAS leaf item
    const TlasLeaf* tlas_leaves;      // parallel to tlas_items: what entering that instance takes
    const InstanceRec* instances;
    const GeometryRec* geometries;
    const MeshRec* meshes;
    const MaterialRec* materials;
    const TexDesc* textures;
    const uint4* texels;              // texel pool in 16-B units, every texture in its own format (TexDesc)
    const float* srgb_lut;            // 256 floats: sRGB byte -> linear, as the host computed it
    const AliasEntry* alias;          // entry 0 = header {count, sum}
    uint32_t alias_count; float alias_sum;   // copy of the header: kernel arguments instead of a dependent load per path
    const LightTri* light_tris;       // alias_count + 1 records
    EnvView env;
    uint32_t tlas_root;               // MAX_UINT when the scene is empty
    uint32_t root_in_blas;            // 1: tlas_root is the root of the merged world BLAS (no TLAS level at all)
    float coord_slack;                // 1.5e-6 x the largest absolute vertex coordinate of the scene, in world space or in the object space of any BLAS (trace.hip step_node)
};
// All visible identity-transform instances are merged into ONE "world BLAS" whose triangle records carry their owning
// instance in TriRec::pad; the TLAS then holds one pseudo-instance for it (InstanceRec flag bit 2) next to the
// transformed instances.  Lanes inside it run with cur_inst == WORLD_INSTANCE.
constexpr uint32_t WORLD_INSTANCE = 0xFFFFFFFEu;
constexpr uint32_t INST_FLAG_VISIBLE = 1u, INST_FLAG_IDENTITY = 2u, INST_FLAG_WORLD = 4u;

// ---- streaming accesses: path state, queues, hit records and finished samples are written once and read once per bounce — 38 GB per 64-launch batch of S1 next to
// a BVH of tens of MB.  Marked non-temporal (the `nt` bit of global_load / global_store) they do not push the BVH out of L2 / Infinity Cache: S1 +0.6 %, 20-launch
// batches +1.1 %, S2 +0.6 %, sky +1.0 % (profiles/r04_nontemporal.txt; -DMSNE_NT=0 for the comparison).  What a kernel reads AGAIN stays temporal: the ray
// direction a two-level scene reloads at every instance entry (non-temporal it cost S2's k_trace_shadow 3 %). ----
#ifndef MSNE_NT
#define MSNE_NT 1
#endif
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t nt_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store(float4* p, float4 v) { if (MSNE_NT) __builtin_nontemporal_store(nt_f4{ v.x, v.y, v.z, v.w }, reinterpret_cast<nt_f4*>(p)); else *p = v; }
__device__ __forceinline__ void nt_store(uint4* p, uint4 v) { if (MSNE_NT) __builtin_nontemporal_store(nt_u4{ v.x, v.y, v.z, v.w }, reinterpret_cast<nt_u4*>(p)); else *p = v; }
__device__ __forceinline__ void nt_store(uint2* p, uint2 v) { if (MSNE_NT) __builtin_nontemporal_store(nt_u2{ v.x, v.y }, reinterpret_cast<nt_u2*>(p)); else *p = v; }
__device__ __forceinline__ float4 nt_load(const float4* p) { if (!MSNE_NT) return *p; const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 nt_load(const uint4* p) { if (!MSNE_NT) return *p; const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 nt_load(const uint2* p) { if (!MSNE_NT) return *p; const nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(p)); return make_uint2(v.x, v.y); }

// ---- wavefront state (one slot per in-flight path; two sets ping-pong between bounces).  Arrays of 16-B records: a wave
// moves each with one coalesced instruction (five loads per path in k_shade instead of twelve 4-B ones) ----
struct PathState {
    float4* ro;     // ray origin xyz | w = flags (bit pattern): the trace kernels fetch a ray with 2 x 16-B loads
    float4* rd;     // ray direction xyz | w unused
    float4* tp;     // throughput rgb | w = pdf of the last BSDF sample
    float4* lr;     // accumulated radiance rgb | w = rng state (bit pattern)
    uint2* sq;      // x = sample slot (s_local * pixels + pixel_local), y = first of the path's light-sample entries in the shadow queue: shq_pack(position in its sub-queue, sub-queue)
    // flags (in ro.w): bits 0..15 bounce count, bit 16 last material delta, bit 17 no ray (finalise only), bit 18 masked,
    // bit 19 light samples pending (env_samples + mesh_samples entries from sq.y, PATH_STRIDE apart), bit 20 dead
};
constexpr uint32_t PATH_FLAG_DELTA = 1u << 16;
constexpr uint32_t PATH_FLAG_ZOMBIE = 1u << 17;   // no ray to trace: the entry only waits for k_shade (to add its light samples and finalise) or is dead / masked
constexpr uint32_t PATH_FLAG_MASKED = 1u << 18;   // slot of a pixel outside the image (edge tiles): ignored by k_shade
constexpr uint32_t PATH_FLAG_NEE = 1u << 19;
constexpr uint32_t PATH_STRIDE_SHIFT = 21;        // bits 21..29: distance between a path's consecutive light-sample entries (the light-sampling paths of its workgroup)
constexpr uint32_t PATH_FLAG_DEAD = 1u << 20;     // ended in k_shade after its queue entry was reserved: ignored by k_shade
constexpr int PATH_STATE_WORDS = 18;

// rec = {instance, triangle slot in SceneView::tris, u bits, v bits}: ONE 16-B store per finished ray; k_shade reads the
// 48-B triangle record (vertices + geometry + primitive) instead of chasing instance → geometry → mesh → indices → positions
struct HitBuf { uint4* rec; };

// One entry per light sample of a path: the shadow ray and the sample's unoccluded contribution.  A workgroup of k_shade
// stores sample k of all its paths together (entry = first + k * stride), env samples before mesh samples.
// `c` is double-buffered by bounce parity (k_shade(b+1) reads bounce b's while writing its own).
struct ShadowQueue {
    float4* o;   // origin xyz | w = tmax (< 0: unused entry — the sample had pdf 0)
    float4* d;   // direction xyz
    float4* c;   // contribution rgb (zeroed by k_trace_shadow when the ray is occluded)
};

// Queue counters of ONE bounce of a batch.  A batch owns an array of them indexed by bounce, zeroed when it starts, so
// nothing has to be rotated or reset between the kernels of a bounce (no bookkeeping launches in the bounce loop):
// k_trace_closest(b) and k_shade(b) consume [b]; k_shade(b) appends to [b + 1]; k_trace_shadow(b) consumes [b + 1]'s shadow half.
//
// The heads k_shade appends to sit on a cache line of their own (QueueHead): in the 32-byte record of rounds 1-4 they shared a line with the neighbouring bounces' counters
// and with the dequeue heads the traversal kernels hammer — every workgroup of k_shade(b) reads the line it starts from while the atomics of the same kernel rewrite it.
// Taking them apart is worth 3-4 % of k_shade (S1 23.0 -> 21.3 ms per 64-launch batch, S2 33.0 -> 31.9: profiles/r05_shade_td.txt).
//
// A queue may be cut into QUEUE_SUBS sub-queues interleaved in tiles of 256 entries: tile t of the queue belongs to sub-queue t % QUEUE_SUBS, entry p of sub-queue k sits at
// queue_slot(p, k), and k_shade appends the survivors of its 256-path chunk c to sub-queue c % QUEUE_SUBS with ONE 64-bit atomic on THAT sub-queue's head.  The reason to want
// it: one device-scope atomic word hands out ~88 reservations per microsecond on this part, and k_shade's first pass makes 162 k of them in 2.2 ms (74 per microsecond) — a
// truncated k_shade that ends right after the reservation spends 4.1 of its 10.5 ms waiting for the one word, and none with eight (profiles/r05_shade_td.txt).  The price is holes
// — the sub-queues end at different lengths, so the last tiles of the shorter ones hold no path; nobody writes or reads them: every consumer derives a queue's extent and which
// entries are live from the heads (queue_dims / queue_live: one compare below the shortest sub-queue's last full tile, a lookup above it).  Chunk c of bounce b is tile c of its
// queue, so sub-queue k of bounce b + 1 is fed by sub-queue k of bounce b alone and can never be longer: extents only shrink.  MEASURED, the whole kernel does not yet run into
// the one-word ceiling: 1 / 4 / 8 / 16 heads give S1 6400 / 6380 / 6325 / 6333 and S2 3846 / 3842 / 3791 / 3805 Mrays/s — the mapping and the hole checks cost what the
// heads save.  The shipped value is ONE sub-queue (a dense queue, as before); the general form stays, compiled as a variant and run by the tests
// (test_sub_queues_render_the_same_film), for the day k_shade is fast enough to need it.
#ifndef MSNE_QUEUE_SUBS
#define MSNE_QUEUE_SUBS 1
#endif
constexpr uint32_t QUEUE_SUBS = MSNE_QUEUE_SUBS, QUEUE_TILE_SHIFT = 8u;   // (a power of two; 1 = one head per queue, as before round 5)
static_assert(QUEUE_SUBS >= 1u && (QUEUE_SUBS & (QUEUE_SUBS - 1u)) == 0u && QUEUE_SUBS <= 256u, "QUEUE_SUBS: a power of two");
// A path names its first shadow-queue entry in ONE word: position in its sub-queue | sub-queue in the top log2(QUEUE_SUBS) bits — with one sub-queue the position keeps all
// 32 bits (round 5 packed 28 | 4 in every build: a 1080p batch of 65 launches at 1 + 1 light samples passes 2^28 entries and read another path's samples; advisor, round 5).
// The host keeps a batch's shadow queue below 2^SHQ_POS_BITS entries (context.hip inflight_budget).
constexpr uint32_t queue_sub_bits(uint32_t n) { return n <= 1u ? 0u : 1u + queue_sub_bits(n >> 1); }
constexpr uint32_t QUEUE_SUB_BITS = queue_sub_bits(QUEUE_SUBS), SHQ_POS_BITS = 32u - QUEUE_SUB_BITS;
MSNE_HD uint32_t shq_pack(uint32_t pos, uint32_t sub) { return QUEUE_SUB_BITS ? (pos | (sub << (SHQ_POS_BITS & 31u))) : pos; }
MSNE_HD uint32_t shq_pos(uint32_t w) { return QUEUE_SUB_BITS ? (w & (0xffffffffu >> QUEUE_SUB_BITS)) : w; }
MSNE_HD uint32_t shq_sub(uint32_t w) { return QUEUE_SUB_BITS ? (w >> (SHQ_POS_BITS & 31u)) : 0u; }
struct alignas(128) QueueHead { uint32_t n_paths, n_shadow; uint32_t pad[30]; };   // (adjacent: ONE 64-bit atomic per workgroup appends to both)
struct alignas(128) BounceCounters {
    QueueHead sub[QUEUE_SUBS];           // path entries | shadow-queue entries of this bounce's queues, per sub-queue
    uint32_t zombies;                    // of the path entries: entries that only wait to be finalised (no ray)
    uint32_t head_closest, head_shadow;  // dequeue heads of k_trace_closest(b) and k_trace_shadow(b - 1)
    uint32_t n_shadow_traced;            // of the shadow entries: entries that held a ray (a light sample with pdf 0 leaves its entry unused), counted by k_trace_shadow
    unsigned long long trunc_pad;        // (what the truncated measurement instantiations of k_shade reserve from: integrator.hip TRUNC)
    uint32_t pad[26];
};
static_assert(sizeof(BounceCounters) == 128 * (QUEUE_SUBS + 1), "BounceCounters layout");
// the shape of one queue, as every consumer derives it from the heads (two registers; the rare entry above dense_end looks its sub-queue's length up in memory)
struct QueueDims { uint32_t extent /* entries, holes included */, dense_end /* every entry below is live */; const BounceCounters* c; };
template <bool SHADOW>
__device__ __forceinline__ QueueDims queue_dims(const BounceCounters* c) {
    QueueDims d; uint32_t tmin = 0xffffffffu, tmax = 0u;
#pragma unroll
    for (uint32_t k = 0; k < QUEUE_SUBS; k++) {
        const uint32_t len = SHADOW ? c->sub[k].n_shadow : c->sub[k].n_paths;
        const uint32_t full = len >> QUEUE_TILE_SHIFT, used = (len + 255u) >> QUEUE_TILE_SHIFT;
        tmin = full < tmin ? full : tmin; tmax = used > tmax ? used : tmax;
    }
    d.extent = (tmax * QUEUE_SUBS) << QUEUE_TILE_SHIFT; d.dense_end = (tmin * QUEUE_SUBS) << QUEUE_TILE_SHIFT; d.c = c;
    return d;
}
__device__ __forceinline__ uint32_t queue_total(const BounceCounters* c, bool shadow) { uint32_t n = 0; for (uint32_t k = 0; k < QUEUE_SUBS; k++) n += shadow ? c->sub[k].n_shadow : c->sub[k].n_paths; return n; }
MSNE_HD uint32_t queue_slot(uint32_t p, uint32_t k) { return ((((p >> QUEUE_TILE_SHIFT) * QUEUE_SUBS) + k) << QUEUE_TILE_SHIFT) | (p & 255u); }
template <bool SHADOW>
__device__ __forceinline__ bool queue_live(const QueueDims& d, uint32_t i) {
    if (i < d.dense_end) return true;
    if (i >= d.extent) return false;
    const uint32_t tile = i >> QUEUE_TILE_SHIFT, k = tile & (QUEUE_SUBS - 1u), p = ((tile / QUEUE_SUBS) << QUEUE_TILE_SHIFT) | (i & 255u);
    return p < (SHADOW ? d.c->sub[k].n_shadow : d.c->sub[k].n_paths);
}
// the lengths of the sub-queues of a DENSE queue of n entries (k_raygen's: queue index = sample slot)
MSNE_HD uint32_t queue_dense_len(uint32_t n, uint32_t k) {
    const uint32_t T = n >> QUEUE_TILE_SHIFT, r = n & 255u;
    return ((T / QUEUE_SUBS + (k < T % QUEUE_SUBS ? 1u : 0u)) << QUEUE_TILE_SHIFT) + (k == T % QUEUE_SUBS ? r : 0u);
}
struct Totals { unsigned long long closest_rays, shadow_rays, samples, pad; };   // since the last MsneResetStats

struct PipelineOpts {   // pipeline.zig:319-327
    uint32_t samples_per_run, max_bounces, env_samples, mesh_samples;
    uint32_t flip_image, indexed_attributes, two_component_normal_texture;
};

// camera constants precomputed on the host with the oracle's expression order (camera.hlsl:14-42)
struct CameraConsts {
    f3 origin, u, v, horizontal, vertical, llc;
    float aperture;
};

// Camera::generateRay (camera.hlsl:14-42) from the precomputed constants; shared by k_raygen and MsnePick
MSNE_HD void camera_generate_ray(const CameraConsts& cam, f2 uv, f2 rand, f3& O, f3& D) {
    const f2 sr = square_to_uniform_disk_concentric(rand);                        // camera.hlsl:31-40
    const f2 rd = F2(cam.aperture * sr.x / 2.0f, cam.aperture * sr.y / 2.0f);
    const f3 defocus = add(scale(cam.u, rd.x), scale(cam.v, rd.y));
    O = add(cam.origin, defocus);
    D = normalize(sub(sub(add(add(cam.llc, scale(cam.horizontal, uv.x)), scale(cam.vertical, uv.y)), defocus), cam.origin));
}

// which pixels this context owns: tile t of the image -> shard t mod shard_count
struct ShardView {
    uint32_t width, height, tile_size, tiles_x, tiles_y;
    uint32_t shard_index, shard_count, local_tiles;   // local tile k = global tile shard_index + k*shard_count
    uint32_t pixels;                                  // local_tiles * tile_size^2 (padded: edge tiles keep masked pixels)
    uint32_t valid_pixels;                            // pixels of this shard that lie inside the image
};

}  // namespace msne
