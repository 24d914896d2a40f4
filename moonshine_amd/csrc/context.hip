// context.hip — implementation of include/moonshine_amd.h: the host side of the hot path.
// Mirrors hydra/hydra.zig:107-558 (the reference's C ABI) and the engine/hrtsystem managers behind it:
// MeshManager.zig (meshes), MaterialManager.zig (materials + TextureManager), Accel.zig (instances, BLAS/TLAS,
// alias table), BackgroundManager.zig (environment), Camera.zig + core/Sensor.zig (lenses, sensors),
// pipeline.zig (specialization constants).  Everything the GPU touches lives in HBM for the life of the
// context; the render loop issues only kernel launches (no per-bounce host synchronisation up to 16 bounces).
#include "../../include/moonshine_amd.h"
#include "msne_device.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <chrono>
#include <mutex>
#include <functional>
#include <string>
#include <vector>

namespace msne {
// kernels' host wrappers (trace.hip, integrator.hip, env.hip, bvh_build.hip)
void launch_trace_closest(hipStream_t, int, bool, const SceneView&, const PathState&, const HitBuf&, BounceCounters*, uint32_t*, uint32_t*, unsigned long long*, uint32_t);
void launch_trace_shadow(hipStream_t, int, bool, const SceneView&, const ShadowQueue&, BounceCounters*, uint32_t*, uint32_t*, unsigned long long*, uint32_t);
void launch_trace_probe(hipStream_t, int, const SceneView&, const float*, uint32_t, int, uint32_t*, uint32_t*, float*, uint32_t*, uint32_t*, uint32_t);
size_t trace_spill_words(int grid);
int trace_blocks_per_cu();
void launch_raygen(hipStream_t, int, const ShardView&, const CameraConsts&, const PipelineOpts&, uint32_t, uint32_t, const PathState&, BounceCounters*);
void launch_shade(hipStream_t, int, const SceneView&, const PipelineOpts&, const PathState&, const HitBuf&, const PathState&, const ShadowQueue&, const float4*, float4*, BounceCounters*, bool, bool);
void launch_account(hipStream_t, const BounceCounters*, uint32_t, Totals*);
void launch_light_tris(hipStream_t, const SceneView&, uint32_t, uint32_t, LightTri*);
void launch_film(hipStream_t, int, const ShardView&, const PipelineOpts&, const float4*, uint32_t, uint32_t, int, int, uint32_t, float4*, float4*);
void launch_unpack_film(hipStream_t, int, const ShardView&, const float4*, uint32_t, uint32_t, size_t, float4*);
bool shade_probe_widths(int, uint32_t&, uint32_t&);
void launch_shade_probe(hipStream_t, const SceneView&, int, const float*, uint32_t, float*);
void launch_env_build(hipStream_t, const float4*, uint32_t, uint32_t, float4*, float*, const uint32_t*, uint32_t, uint32_t);
void launch_env_quads(hipStream_t, const float*, const uint32_t*, float4*, const uint32_t*, uint32_t, uint32_t);
struct BlasGeo { const float* positions; const uint32_t* indices; uint32_t tri_offset, tri_count, geo, inst; const float* normals; const float* texcoords; uint32_t indexed, attr_count; };
struct BuildScratch;   // per-context build buffers (bvh_build.hip)
BuildScratch* bvh_scratch_create();
void bvh_scratch_set_fast(BuildScratch*, bool);
void bvh_scratch_destroy(BuildScratch*);
void bvh_scratch_release(BuildScratch*);
size_t bvh_scratch_capacity(const BuildScratch*);
bool bvh_build_blas_batch(BuildScratch*, hipStream_t, const std::vector<BlasGeo>&, const std::vector<uint32_t>&, Node8*, uint32_t*, uint32_t, TriRec*, TriRot*, TriAttr*, uint32_t*, uint32_t*, uint32_t*, float*);
struct TlasInst { float T[12]; float blas_box[6]; uint32_t mesh_begin, mesh_end, exact; float cull_pad; };   // (bvh_build.hip)
struct TlasMesh { const float* positions; uint32_t count, pad; };
bool bvh_build_tlas(BuildScratch*, hipStream_t, const TlasInst*, const uint32_t*, uint32_t, const TlasMesh*, uint32_t, Node8*, uint32_t*, uint32_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, float4*);
void bvh_tlas_links(hipStream_t, const Node8*, uint32_t node_begin, uint32_t node_end, uint32_t item_begin, uint2* node_parent, uint2* item_parent, uint32_t root);
void bvh_tlas_leaves(hipStream_t, const uint32_t* items, const InstanceRec* instances, const float4* spheres, uint32_t n, TlasLeaf* out);
void bvh_scratch_release_arena(BuildScratch*);
size_t bvh_scratch_arena_bytes(const BuildScratch*);
bool bvh_refit_tlas(BuildScratch*, hipStream_t, const TlasInst*, const TlasMesh*, uint32_t, const uint32_t* edit_items, uint32_t n_edits, Node8*, uint32_t node_begin, uint32_t n_nodes, uint32_t item_begin, uint32_t n_items, const uint2*, const uint2*, const uint32_t* edit_ids, float4* spheres);
}  // namespace msne

using namespace msne;

static thread_local std::string g_create_error;

#define CHECK_HIP(ctx, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (ctx)->fail(std::string("HIP: ") + hipGetErrorString(e_) + " (" #x ")"); return false; } } while (0)

namespace {

template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    bool alloc(size_t count) {
        release(); if (count == 0) count = 1;
        if (hipMalloc(&p, count * sizeof(T)) != hipSuccess) { p = nullptr; return false; }
        static const bool poison = debug_poison();   // tests: device memory starts as garbage (0xCD), as recycled memory does in a long-lived process
        if (poison && (hipMemset(p, 0xCD, count * sizeof(T)) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) { (void)hipFree(p); p = nullptr; return false; }   // (the library's streams do not wait for the null stream)
        n = count; return true;
    }
    bool ensure(size_t count) { return count <= n && p ? true : alloc(count); }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    ~DevBuf() { release(); }
    DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
};

struct TextureH { uint32_t w, h, format; std::vector<uint8_t> raw; float first[4]; size_t offset = 0; bool on_device = false; };   // raw: the texels in their own format until they are on the device, then released; offset in 16-B units
struct MeshH {
    DevBuf<float> positions, normals, texcoords; DevBuf<uint32_t> indices;
    std::vector<float> h_positions; std::vector<uint32_t> h_indices;   // host copies for the alias-table areas (Accel.zig:503-519)
    uint32_t position_count = 0, attribute_count = 0, index_count = 0, max_index = 0; bool has_normals = false, has_texcoords = false;
};
struct InstanceH { m34 transform; bool visible; std::vector<GeometryRec> geos; };
struct BlasInfo { uint32_t root; float box[6]; uint32_t tris; };
struct MaterialUpdate { bool has[6] = { false, false, false, false, false, false }; uint32_t tex[5] = { 0, 0, 0, 0, 0 }; float ior = 0.0f; };

struct SensorH {
    Extent2D extent{}; uint32_t sample_count = 0;
    ShardView shard{};
    DevBuf<float4> film_packed, film_full, color;
    float4* host = nullptr;   // pinned, row-major float4[w*h]
};

struct PathBuffers {
    DevBuf<float> f;      // ro, rd, tp, lr (float4 each) + sq (uint2)
    size_t cap = 0;
    PathState view() const {
        PathState s; float* b = f.p; size_t c = cap;
        s.ro = reinterpret_cast<float4*>(b); s.rd = reinterpret_cast<float4*>(b + 4 * c); s.tp = reinterpret_cast<float4*>(b + 8 * c);
        s.lr = reinterpret_cast<float4*>(b + 12 * c); s.sq = reinterpret_cast<uint2*>(b + 16 * c);
        return s;
    }
    bool ensure(size_t c) { if (c <= cap) return true; cap = 0; if (!f.alloc(18 * c)) return false; cap = c; return true; }
};

}  // namespace

struct HdMoonshine {
    std::mutex mutex;                 // hydra.zig:76-78: every call is serialised
    int device = 0;
    hipStream_t stream = nullptr;
    std::string last_error;
    MsneConfig cfg{};
    PipelineOpts opts{};

    std::vector<TextureH> textures; bool textures_dirty = true;
    std::vector<MeshH*> meshes;
    std::vector<MaterialRec> materials; bool materials_dirty = true;
    std::map<uint32_t, MaterialUpdate> material_updates;   // deferred edits (hydra.zig:152-223)
    std::vector<InstanceH> instances; bool accel_dirty = true; bool tlas_only_dirty = false;
    std::vector<std::pair<uint32_t, uint32_t>> geometry_edits;   // {instance, geometry} whose material changed since the tables were uploaded (Accel.zig:609-628)
    std::vector<Lens> lenses;
    std::vector<SensorH*> sensors;

    // device scene tables
    DevBuf<uint4> d_texels; DevBuf<float> d_srgb; DevBuf<TexDesc> d_texdesc; size_t texels_end = 0;   // texel pool (16-B units, each texture in its own format); texels_end: what is already on the device
    DevBuf<MaterialRec> d_materials;
    DevBuf<MeshRec> d_meshes;
    DevBuf<GeometryRec> d_geometries;
    DevBuf<InstanceRec> d_instances;
    DevBuf<AliasEntry> d_alias;
    DevBuf<LightTri> d_light_tris; bool lights_dirty = true; uint32_t lights_indexed = 0;   // gathered light triangles (rebuilt with the alias table / attribute mode)
    DevBuf<Node8> d_nodes; DevBuf<TriRec> d_tris; DevBuf<TriRot> d_tri_rot; DevBuf<TriAttr> d_tri_attrs; DevBuf<uint32_t> d_tlas_items, d_item_src; DevBuf<TlasLeaf> d_tlas_leaves; DevBuf<float4> d_inst_spheres;   // per instance (+ the world pseudo-instance): world-space bounding sphere
    bool blas_indexed = true;                             // the attribute mode the TriAttr records of the cached BLASes were gathered with
    DevBuf<uint32_t> d_build_counters;    // [0] node count, [1] tri count, [2] tlas item count
    uint32_t blas_nodes_end = 0, blas_tris_end = 0;
    std::map<std::vector<uint32_t>, BlasInfo> blas_cache;
    std::vector<uint32_t> world_blas_key;                 // key of the ONE world BLAS kept in blas_cache (empty: none)
    uint32_t dead_tris = 0;                               // triangles of evicted world BLASes still lying in the pools (reclaimed by a pool reset)
    BuildScratch* build_scratch = nullptr;                // this context's BVH build buffers: nothing is shared between contexts
    uint32_t tlas_root = MAX_UINT, root_in_blas = 0, n_tlas_items = 0;
    float coord_radius = 0.0f;   // bound on the absolute vertex coordinates of the built scene, in world space and in every BLAS's object space (SceneView::coord_slack)
    // Ray origins and the instances' culling volumes (instance_cull_pad: the |o| term).  Secondary rays start on surfaces — inside coord_radius; PRIMARY rays start where the
    // caller puts them (intersection.hlsl:20: TraceRay takes any origin), and the host knows where before it launches anything: the lens of a render or a pick, the origins
    // handed to MsneTraceRays.  `origin_needed` is that call's largest origin coordinate; the TLAS in use was baked for `baked_origin_reach` (16 x coord_radius at least —
    // ordinary cameras never get here); an origin beyond it re-bakes every pad and sphere for twice what was asked (`origin_reach_floor`, kept: a camera that keeps backing
    // away re-bakes at every doubling, not at every frame).  A TLAS-only rebuild: the BLASes are cached, 17 ms for 100 000 instances.
    float origin_needed = 0.0f, origin_reach_floor = 0.0f, baked_origin_reach = 0.0f;
    float origin_reach_wanted() {
        if (origin_needed > 16.0f * coord_radius && origin_needed > origin_reach_floor) origin_reach_floor = 2.0f * origin_needed;
        return std::min(std::max(16.0f * coord_radius, origin_reach_floor), 3.0e38f);   // (a scene with a coordinate beyond 2e37: the product must stay finite, or every instance's slack is infinite)
    }
    void need_origin(float x, float y, float z, float extra = 0.0f) {   // (NaN and infinite origins hit nothing in any space: they ask for nothing)
        const float m = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) + fabsf(extra);
        if (m < 1.5e38f && m > origin_needed) origin_needed = m;
    }
    // in-place TLAS update (Accel.zig:567-601): transform edits since the last build, and what a refit needs of that build
    std::vector<uint32_t> transform_edits;
    std::vector<char> built_in_world; std::vector<uint32_t> item_of_instance;   // per instance, as of the last rebuild (MAX_UINT: not in the TLAS)
    DevBuf<uint2> d_tlas_node_parent, d_tlas_item_parent; uint32_t tlas_node_begin = 0, tlas_node_end = 0, tlas_item_begin = 0;
    std::vector<InstanceRec> h_irec;
    uint64_t n_rebuilds = 0, n_tlas_updates = 0; uint32_t refits_since_rebuild = 0;
    bool fast_builds = false;                      // MsneSetBuildQuality
    bool refit_tlas();
    std::vector<AliasEntry> h_alias;
    // environment
    DevBuf<float4> d_env_rgb, d_env_quads; DevBuf<float> d_env_lum; EnvView env{};
    // wavefront
    // A pipe = one independent wavefront pipeline (state ping-pong, hit/shadow queues, counters, two streams).  A batch of
    // launches is split over up to n_pipes pipes that run concurrently: while one pipe is in the thin tail of a bounce
    // (a few long rays, most CUs idle) the others' bulk work fills the machine.
    struct Pipe {
        PathBuffers paths[2];
        DevBuf<uint32_t> hit_u, spill, spill2; DevBuf<float> shq_f; uint32_t shq_samples = 0;   // shadow queue: o, d, c x 2, each samples * cap float4
        DevBuf<BounceCounters> counters;   // one per bounce of the batch in flight (+1)
        DevBuf<Totals> totals;
        hipStream_t s0 = nullptr, s1 = nullptr;   // s1: k_trace_shadow, overlapped with the next bounce's k_trace_closest
        size_t cap = 0;
    };
    static constexpr int MAX_PIPES = 4;
    Pipe pipes[MAX_PIPES];
    // k_trace_shadow(b) may overlap k_trace_closest(b+1) on a second stream.  Both kernels are VALU-issue bound, so while their queues are
    // long the overlap buys nothing (S1 x 64 launches: 108.5 ms overlapped, 108.8 ms in stream order) and only blurs per-kernel timings;
    // it pays in the thin tails of a batch, where neither kernel fills the chip.  $MSNE_SERIAL: 1 = always in stream order, 0 = always
    // overlapped, unset = in stream order for the first `serial_bounces` bounces of batches of at least `serial_min_paths` paths.
    int serial_mode = -1, serial_saved = -2; uint32_t serial_bounces = 4; size_t serial_min_paths = 96u << 20;   // (overlap is worth 2.5 % on a 2-way shard = 67 M paths, 3.6 % at 20 launches = 41 M, 6 % on an 8-way shard)
    int n_pipes = 1;                               // $MSNE_PIPES (measured on S1: more pipes never won — bigger batches beat overlapped smaller ones)
    size_t single_pipe_paths = 48u << 20;          // batches at least this large run on one pipe (tails are negligible there)
    DevBuf<float4> d_lbuf;
    size_t lbuf_cap = 0;
    DevBuf<uint32_t> d_overflow; DevBuf<unsigned long long> d_trace_stats;
    int trace_grid = 1024, shade_grid = 2048, shade_k_grid = 2048;
    uint32_t refill = 16;              // traversal: idle lanes (of 64) at which a wave refills from the ray queue; $MSNE_REFILL
    size_t max_inflight = 160u << 20;  // most paths traced concurrently (288 B of wavefront state each at one env + one mesh light sample, allocated on demand); $MSNE_MAX_INFLIGHT
    // ... and what a batch may hold of it: a path names its first shadow-queue entry by a position of SHQ_POS_BITS bits (msne_device.h shq_pack), a pipe's shadow queue
    // holds (env + mesh samples) entries per path of its capacity (the paths rounded up to a round of sub-queue tiles: ensure_wavefront), so capacity x samples stays below
    // 2^SHQ_POS_BITS.  One sub-queue, 1 + 1 samples: 2 Gi paths, far above the default; eight sub-queues and 4 + 4 samples: 64 Mi — batches get smaller, nothing wraps.
    size_t inflight_budget() const {
        const size_t ns = std::max(1u, opts.env_samples + opts.mesh_samples), slack = 256 + (size_t)QUEUE_SUBS * 256;
        const size_t most = ((size_t)1 << SHQ_POS_BITS) / ns;
        return std::max<size_t>(1, std::min(max_inflight, most > 2 * slack ? most - slack : most / 2));
    }
    // statistics
    MsneStats stats{};
    bool profile = false, trace_stats = false;
    std::vector<hipEvent_t> events; size_t events_used = 0;
    struct Span { size_t a, b; int kind; };
    std::vector<Span> spans;
    std::vector<std::pair<int, float>> launch_times;   // the last profiled render's trace / shade launches in issue order: {kind, ms} (MsneGetLaunchTimes)

    void fail(const std::string& m) { last_error = m; if (getenv("MSNE_VERBOSE")) fprintf(stderr, "moonshine_amd: %s\n", m.c_str()); }
    void clear_all_sensors() { for (auto* s : sensors) s->sample_count = 0; }   // Camera.clearAllSensors Camera.zig:73-77
    bool bind() { if (hipSetDevice(device) != hipSuccess) { fail("hipSetDevice failed"); return false; } return true; }

    hipEvent_t next_event() {
        if (events_used == events.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; events.push_back(e); }
        return events[events_used++];
    }

    bool upload_textures();
    bool upload_materials();
    bool rebuild_accel();
    bool ensure_scene();
    bool ensure_wavefront(size_t paths_per_pipe, size_t slots, int npipes);
    SceneView scene_view() const;
    bool set_background(const float* rgba, Extent2D e);
    bool render(uint32_t sensor, uint32_t lens, uint32_t launches, bool readback);
    bool readback(SensorH* s);
    // the traversal kernels raise a flag when a ray needs more than 128 stack entries; read and re-arm it after every render / probe
    bool check_overflow() {
        uint32_t overflow = 0;
        CHECK_HIP(this, hipMemcpy(&overflow, d_overflow.p, 4, hipMemcpyDeviceToHost));
        if (!overflow) return true;
        CHECK_HIP(this, hipMemset(d_overflow.p, 0, 4));
        fail("traversal stack overflow");
        return false;
    }
    ~HdMoonshine();
};

// ---------------- textures ----------------
namespace msne_host { void parallel_for(uint32_t n, const std::function<void(uint32_t)>& job); }   // host/exr.cpp: the host threads this process may use

static const float* srgb_lut() {
    struct Lut { float v[256]; Lut() { for (int i = 0; i < 256; i++) { const double x = i / 255.0; v[i] = (float)(x <= 0.04045 ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4)); } } };
    static const Lut lut;   // (initialised once, thread-safely: contexts on different threads create textures concurrently)
    return lut.v;
}
static int64_t add_texture(HdMoonshine* c, const void* bytes, uint32_t w, uint32_t h, int fmt) {
    if (!bytes || w == 0 || h == 0) { c->fail("texture: bad arguments"); return -1; }
    if (fmt < 0 || fmt > MSNE_FORMAT_R16G16B16A16_SFLOAT) { c->fail("texture: unknown format"); return -1; }
    // the texels are kept as they come (MaterialManager.zig:351-390 uploads a texture in its own vk.Format): a lookup decodes what it touches
    TextureH t; t.w = w; t.h = h; t.format = (uint32_t)fmt;
    const size_t nbytes = (size_t)w * h * tex_bytes_per_texel((uint32_t)fmt);
    t.raw.assign((const uint8_t*)bytes, (const uint8_t*)bytes + nbytes);
    t.raw.resize((nbytes + 15) & ~(size_t)15, 0);   // textures start on 16-B boundaries of the pool
    const float4 f0 = texel_decode(t.raw.data(), t.format, 0, srgb_lut());
    t.first[0] = f0.x; t.first[1] = f0.y; t.first[2] = f0.z; t.first[3] = f0.w;
    c->textures.push_back(std::move(t));
    c->textures_dirty = true;
    return (int64_t)c->textures.size() - 1;
}

// Textures are append-only (the reference never frees one either, MaterialManager.zig): the texel pool grows like the triangle pools — what is already
// on the device is moved device-to-device, only new textures cross PCIe, and their host copies are released once they are uploaded, so a scene with
// gigabytes of texels keeps them once, in HBM, in the source's own format (4 B per sRGB texel, 1 B per metalness / roughness texel, ...).
bool HdMoonshine::upload_textures() {
    size_t end = texels_end, total = texels_end;   // in 16-B units
    for (auto& t : textures) if (!t.on_device) { t.offset = total; total += t.raw.size() / 16; }
    if (total > 0xFFFFFFFFull) { fail("more than 64 GiB of texels"); return false; }
    if (total > d_texels.n || !d_texels.p) {
        DevBuf<uint4> nt; if (!nt.alloc(total + total / 4 + 16)) { fail("out of device memory (textures)"); return false; }
        if (end) CHECK_HIP(this, hipMemcpyAsync(nt.p, d_texels.p, end * sizeof(uint4), hipMemcpyDeviceToDevice, stream));
        CHECK_HIP(this, hipStreamSynchronize(stream));
        std::swap(nt.p, d_texels.p); std::swap(nt.n, d_texels.n);
    }
    if (!d_srgb.p) {
        if (!d_srgb.alloc(256)) { fail("out of device memory (textures)"); return false; }
        CHECK_HIP(this, hipMemcpyAsync(d_srgb.p, srgb_lut(), 256 * sizeof(float), hipMemcpyHostToDevice, stream));
    }
    std::vector<TexDesc> desc(textures.size());
    for (size_t i = 0; i < textures.size(); i++) {
        TextureH& t = textures[i];
        desc[i] = TexDesc{ (uint32_t)t.offset, t.w, t.h, t.format, make_float4(t.first[0], t.first[1], t.first[2], t.first[3]) };
        if (!t.on_device) CHECK_HIP(this, hipMemcpyAsync(d_texels.p + t.offset, t.raw.data(), t.raw.size(), hipMemcpyHostToDevice, stream));
    }
    if (!d_texdesc.alloc(desc.size())) { fail("out of device memory (textures)"); return false; }
    if (!desc.empty()) CHECK_HIP(this, hipMemcpyAsync(d_texdesc.p, desc.data(), desc.size() * sizeof(TexDesc), hipMemcpyHostToDevice, stream));
    CHECK_HIP(this, hipStreamSynchronize(stream));
    for (auto& t : textures) if (!t.on_device) { t.on_device = true; std::vector<uint8_t>().swap(t.raw); }
    texels_end = total;
    textures_dirty = false; lights_dirty = true;   // (the gathered light triangles hold texture descriptors)
    return true;
}

bool HdMoonshine::upload_materials() {
    // flush deferred edits (hydra.zig:152-223)
    for (auto& kv : material_updates) {   // the Hydra setters return nothing: a bad handle surfaces here, at the Render that would have used it
        const MaterialUpdate& u0 = kv.second;
        bool bad = kv.first >= materials.size();
        for (int f = 0; f < 5; f++) if (u0.has[f] && u0.tex[f] >= textures.size()) bad = true;
        if (bad) {
            const uint32_t which = kv.first;
            material_updates.clear();
            fail("material edit: unknown material or texture handle (material " + std::to_string(which) + "); the edit was dropped");
            return false;
        }
    }
    for (auto& kv : material_updates) {
        MaterialRec& m = materials[kv.first]; const MaterialUpdate& u = kv.second;
        if (u.has[0]) m.normal = u.tex[0];
        if (u.has[1]) m.emissive = u.tex[1];
        if (u.has[2]) m.color = u.tex[2];
        if (u.has[3]) m.metalness = u.tex[3];
        if (u.has[4]) m.roughness = u.tex[4];
        if (u.has[5]) m.ior = u.ior;
    }
    material_updates.clear();
    if (!d_materials.alloc(materials.size())) { fail("out of device memory (materials)"); return false; }
    if (!materials.empty()) CHECK_HIP(this, hipMemcpyAsync(d_materials.p, materials.data(), materials.size() * sizeof(MaterialRec), hipMemcpyHostToDevice, stream));
    CHECK_HIP(this, hipStreamSynchronize(stream));
    materials_dirty = false; lights_dirty = true;   // (... of their materials' emissive textures)
    return true;
}

// ---------------- acceleration structure + alias table (Accel.zig:312-563) ----------------
static void vose_alias(const std::vector<float>& w, AliasEntry* entries, float& sum_out) {   // alias_table.zig:25-92
    const uint32_t n = (uint32_t)w.size();
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; i++) sum += w[i];
    uint32_t less_head = MAX_UINT, more_head = MAX_UINT;
    for (uint32_t i = 0; i < n; i++) {
        const float adj = (w[i] * (float)n) / sum;
        entries[i].select = adj;
        if (adj < 1.0f) { entries[i].alias = less_head; less_head = i; } else { entries[i].alias = more_head; more_head = i; }
    }
    while (less_head != MAX_UINT && more_head != MAX_UINT) {
        const uint32_t less = less_head; less_head = entries[less].alias;
        const uint32_t more = more_head; more_head = entries[more].alias;
        entries[less].alias = more;
        entries[more].select = (entries[more].select + entries[less].select) - 1.0f;
        if (entries[more].select < 1.0f) { entries[more].alias = less_head; less_head = more; } else { entries[more].alias = more_head; more_head = more; }
    }
    while (less_head != MAX_UINT) { const uint32_t less = less_head; less_head = entries[less].alias; entries[less].select = 1.0f; }
    sum_out = sum;
}

static bool is_identity(const m34& m) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) if (m.m[r][c] != ((r == c) ? 1.0f : 0.0f)) return false;
    return true;
}

// An instance under a transform with an infinite or NaN entry cannot be hit — its inverse, and with it every ray taken into its space, has a NaN in it, and the triangle
// test accepts nothing that is not a number (the oracle enters such an instance and finds nothing) — so it is left out of the TLAS like a hidden one: boxes of +-3e38
// around it would take every other instance's box of the same TLAS node down to one or two quanta of a grid 1e38 wide.
static bool finite_transform(const m34& m) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) if (!(fabsf(m.m[r][c]) < 3.0e38f)) return false;
    return true;
}

// largest absolute coordinate an instance's vertices can have: in the BLAS's own space (its root box) and under the transform (|T| applied to the box's reach)
static float coord_reach(const m34& T, const float box[6]) {
    float a[3], r = 0.0f;
    for (int j = 0; j < 3; j++) { a[j] = std::max(fabsf(box[j]), fabsf(box[j + 3])); r = std::max(r, a[j]); }
    for (int k = 0; k < 3; k++) r = std::max(r, fabsf(T.m[k][0]) * a[0] + fabsf(T.m[k][1]) * a[1] + fabsf(T.m[k][2]) * a[2] + fabsf(T.m[k][3]));
    return r < 3.0e38f ? r : 3.0e38f;   // (NaN or infinite: no culling against the best hit at all)
}

// How far a world-space ray can pass from an instance's world-space vertices and still hit one of its triangles: the hit is decided in instance space on
// fl(W o + w) + s fl(W d) — W, w the ROUNDED inverse (m34_inverse_affine, f32) — and T (W p + w) + t is not p.  For a point p of the ray and its twin q in instance space
//     |T q + t - p| <= |T W - I| |p| + |T w + t| + |T| g (|W| (2 |o| + |p|) + |w|),        g = four roundings of a dot product, doubled,
// plus g (|T| |v| + |t|) for the rounding of the transformed vertices themselves; with |p| <= the instance's reach (+ the slack) and |o| <= origin_reach, in double.
// Every world-space volume an instance is culled by (its TLAS leaf box and, through it, the boxes above; its bounding sphere) is grown by this much
// (bvh_build.hip k_instance_boxes).  Well-conditioned transforms: a few ulps of the coordinates.  A shear between scales 1e6 apart, or an instance a few ulps of its own
// coordinates wide: as large as the instance — nothing is culled there, which is the contract (the oracle's search without boxes, tests/test_oracle.py).  Ray origins
// farther out than origin_reach are the host's business: it knows every primary origin before it launches (HdMoonshine::need_origin) and re-bakes; DESIGN.md section 2.
static float instance_cull_pad(const m34& T, const float box[6], float origin_reach) {
    if (is_identity(T)) return 0.0f;   // (the traversal does not transform at all: trace.hip `ident`)
    const m34 W = m34_inverse_affine(T);
    const double g = 8.0 / 16777216.0;
    double reach = 0.0, fixed = 0.0, resid = 0.0, ampl = 0.0;
    for (int i = 0; i < 3; i++) {
        double r_row = 0.0, a_row = 0.0, tau = T.m[i][3], w_row = 0.0, world = fabs((double)T.m[i][3]);
        for (int j = 0; j < 3; j++) {
            double r = i == j ? -1.0 : 0.0, a = 0.0;
            for (int k = 0; k < 3; k++) { r += (double)T.m[i][k] * W.m[k][j]; a += fabs((double)T.m[i][k]) * fabs((double)W.m[k][j]); }
            r_row += fabs(r); a_row += a;
            tau += (double)T.m[i][j] * W.m[j][3]; w_row += fabs((double)T.m[i][j]) * fabs((double)W.m[j][3]);
            world += fabs((double)T.m[i][j]) * std::max(fabs((double)box[j]), fabs((double)box[j + 3]));
        }
        reach = std::max(reach, world); resid = std::max(resid, r_row); ampl = std::max(ampl, a_row);
        fixed = std::max(fixed, fabs(tau) + g * w_row + g * world);
    }
    fixed += 2.0 * g * ampl * (double)origin_reach;
    double e = fixed + (resid + g * ampl) * reach;
    e = fixed + (resid + g * ampl) * (reach + e);
    e = 1.5 * (fixed + (resid + g * ampl) * (reach + e));
    return (e == e && e < 1e37) ? (float)e * 1.000001f : 3.0e38f;   // (not finite: boxes and a sphere that nothing misses)
}

bool HdMoonshine::rebuild_accel() {
    // $MSNE_BUILD_TIMING: host wall time of the phases of a rebuild on stderr
    static const bool timing = getenv("MSNE_BUILD_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    std::string t_report;
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        char b[96]; snprintf(b, sizeof b, " %s %.2f ms;", what, std::chrono::duration<double, std::milli>(t - t_prev).count()); t_report += b; t_prev = t;
    };
    // flat geometry table + per-instance offsets (Accel.zig:362-412)
    std::vector<GeometryRec> geos; std::vector<uint32_t> geo_offset(instances.size());
    for (size_t i = 0; i < instances.size(); i++) { geo_offset[i] = (uint32_t)geos.size(); for (auto& g : instances[i].geos) geos.push_back(g); }
    for (auto& g : geos) if (g.mesh < meshes.size()) g.sampled = (g.sampled ? GEO_SAMPLED : 0u) | (meshes[g.mesh]->has_texcoords ? GEO_HAS_TEXCOORDS : 0u) | (meshes[g.mesh]->has_normals ? GEO_HAS_NORMALS : 0u);
    for (auto& g : geos) {
        if (g.mesh >= meshes.size()) { fail("instance references an unknown mesh"); return false; }
        if (g.material >= materials.size()) { fail("instance references an unknown material"); return false; }
    }
    std::vector<MeshRec> mrec(meshes.size());
    for (size_t i = 0; i < meshes.size(); i++) mrec[i] = MeshRec{ meshes[i]->positions.p, meshes[i]->has_texcoords ? meshes[i]->texcoords.p : nullptr, meshes[i]->has_normals ? meshes[i]->normals.p : nullptr, meshes[i]->indices.p };
    if (!d_meshes.alloc(mrec.size()) || !d_geometries.alloc(geos.size())) { fail("out of device memory (tables)"); return false; }
    if (!mrec.empty()) CHECK_HIP(this, hipMemcpyAsync(d_meshes.p, mrec.data(), mrec.size() * sizeof(MeshRec), hipMemcpyHostToDevice, stream));
    if (!geos.empty()) CHECK_HIP(this, hipMemcpyAsync(d_geometries.p, geos.data(), geos.size() * sizeof(GeometryRec), hipMemcpyHostToDevice, stream));

    // BLAS per unique mesh list (Accel.zig:315-343), cached across rebuilds — except that all visible
    // identity-transform instances are merged into ONE world-space BLAS (msne_device.h WORLD_INSTANCE): static
    // geometry then needs no TLAS hop, no ray transform and no per-instance root visits.
    const size_t N = instances.size();
    std::vector<std::vector<uint32_t>> keys(N);
    std::vector<char> in_world(N, 0);
    std::vector<uint32_t> world_key;
    for (size_t i = 0; i < N; i++) {
        uint32_t ntri = 0;
        for (auto& g : instances[i].geos) { keys[i].push_back(g.mesh); ntri += meshes[g.mesh]->index_count; }
        if (instances[i].visible && ntri && is_identity(instances[i].transform)) {
            in_world[i] = 1;
            world_key.push_back(0xFFFFFFFFu); world_key.push_back((uint32_t)i);
            world_key.insert(world_key.end(), keys[i].begin(), keys[i].end());
        }
    }
    lap("tables + keys");
    // At most one world BLAS is kept: when the set of visible identity instances changes (Hydra visibility / transform edits,
    // hydra.zig:495-513) the previous one is evicted, and once evicted BLASes make up more than half of the pools the pools are
    // reset and everything still referenced is rebuilt — device memory stays within 2x of what the scene needs however long
    // an interactive session edits instances.
    // The TriAttr records hold the attributes as THIS pipeline reads them (by vertex index or by corner): a change of mode rebuilds
    if (blas_indexed != (opts.indexed_attributes != 0)) {
        blas_cache.clear(); world_blas_key.clear(); blas_nodes_end = 0; blas_tris_end = 0; dead_tris = 0;
        blas_indexed = opts.indexed_attributes != 0;
    }
    auto blas_geo = [&](const MeshH* m, uint32_t off, uint32_t g, uint32_t inst) {
        return BlasGeo{ m->positions.p, m->indices.p, off, m->index_count, g, inst, m->has_normals ? m->normals.p : nullptr, m->has_texcoords ? m->texcoords.p : nullptr,
                        blas_indexed ? 1u : 0u, m->attribute_count };
    };
    if (world_key != world_blas_key) {
        auto old = blas_cache.find(world_blas_key);
        if (!world_blas_key.empty() && old != blas_cache.end()) { dead_tris += old->second.tris; blas_cache.erase(old); }
        world_blas_key = world_key;
        if ((size_t)dead_tris * 2 > (size_t)blas_tris_end) { blas_cache.clear(); blas_nodes_end = 0; blas_tris_end = 0; dead_tris = 0; }
    }
    size_t new_tris = 0;
    {
        std::map<std::vector<uint32_t>, bool> seen;
        for (size_t i = 0; i < N; i++) {
            if (in_world[i]) continue;
            if (!blas_cache.count(keys[i]) && !seen.count(keys[i])) { seen[keys[i]] = true; for (uint32_t m : keys[i]) new_tris += meshes[m]->index_count; }
        }
        if (!world_key.empty() && !blas_cache.count(world_key))
            for (size_t i = 0; i < N; i++) if (in_world[i]) for (uint32_t m : keys[i]) new_tris += meshes[m]->index_count;
    }
    if (!build_scratch && !(build_scratch = bvh_scratch_create())) { fail("out of host memory"); return false; }
    bvh_scratch_set_fast(build_scratch, fast_builds);
    if (!d_build_counters.p) { if (!d_build_counters.alloc(4)) { fail("out of device memory"); return false; } CHECK_HIP(this, hipMemsetAsync(d_build_counters.p, 0, 16, stream)); }
    const size_t need_tris = (size_t)blas_tris_end + new_tris;
    const size_t need_nodes = (size_t)blas_nodes_end + new_tris + 2 * N + 64;
    if (need_tris > d_tris.n || !d_tris.p) {
        DevBuf<TriRec> nt; DevBuf<TriRot> nr;
        if (!nt.alloc(need_tris + need_tris / 4 + 16) || !nr.alloc(need_tris + need_tris / 4 + 16)) { fail("out of device memory (triangles)"); return false; }
        if (blas_tris_end) CHECK_HIP(this, hipMemcpyAsync(nt.p, d_tris.p, (size_t)blas_tris_end * sizeof(TriRec), hipMemcpyDeviceToDevice, stream));
        if (blas_tris_end) CHECK_HIP(this, hipMemcpyAsync(nr.p, d_tri_rot.p, (size_t)blas_tris_end * sizeof(TriRot), hipMemcpyDeviceToDevice, stream));
        CHECK_HIP(this, hipStreamSynchronize(stream));
        std::swap(nt.p, d_tris.p); std::swap(nt.n, d_tris.n); std::swap(nr.p, d_tri_rot.p); std::swap(nr.n, d_tri_rot.n);
    }
    // TriAttr records ride in the same slots as the TriRecs, allocated once any referenced mesh carries normals or texcoords
    bool any_attrs = false;
    for (size_t i = 0; i < N && !any_attrs; i++) for (uint32_t m : keys[i]) if (meshes[m]->has_normals || meshes[m]->has_texcoords) { any_attrs = true; break; }
    if (any_attrs && d_tri_attrs.n < d_tris.n) {
        DevBuf<TriAttr> na; if (!na.alloc(d_tris.n)) { fail("out of device memory (triangle attributes)"); return false; }
        if (blas_tris_end && d_tri_attrs.p) CHECK_HIP(this, hipMemcpyAsync(na.p, d_tri_attrs.p, std::min((size_t)blas_tris_end, d_tri_attrs.n) * sizeof(TriAttr), hipMemcpyDeviceToDevice, stream));
        CHECK_HIP(this, hipStreamSynchronize(stream));
        std::swap(na.p, d_tri_attrs.p); std::swap(na.n, d_tri_attrs.n);
    }
    if (need_nodes > d_nodes.n || !d_nodes.p) {
        DevBuf<Node8> nn; if (!nn.alloc(need_nodes + need_nodes / 4 + 16)) { fail("out of device memory (nodes)"); return false; }
        if (blas_nodes_end) CHECK_HIP(this, hipMemcpyAsync(nn.p, d_nodes.p, (size_t)blas_nodes_end * sizeof(Node8), hipMemcpyDeviceToDevice, stream));
        CHECK_HIP(this, hipStreamSynchronize(stream));
        std::swap(nn.p, d_nodes.p); std::swap(nn.n, d_nodes.n);
    }
    if (!d_item_src.ensure(std::max(d_tris.n, N + 2)) || !d_tlas_items.ensure(N + 2) || !d_tlas_leaves.ensure(N + 2) || !d_inst_spheres.ensure(N + 2)) { fail("out of device memory (items)"); return false; }
    // counters: node count resumes after the BLAS region (the previous TLAS is discarded)
    { uint32_t c[4] = { blas_nodes_end, blas_tris_end, 0u, 0u }; CHECK_HIP(this, hipMemcpyAsync(d_build_counters.p, c, 16, hipMemcpyHostToDevice, stream)); CHECK_HIP(this, hipStreamSynchronize(stream)); }
    lap("pools");
    {   // every BLAS this rebuild needs — one per unique mesh list not yet cached, and the world BLAS — goes through ONE pass of the builder
        std::vector<BlasGeo> bg; std::vector<uint32_t> job_first{ 0u }; std::vector<std::vector<uint32_t>> job_key;
        uint32_t off = 0;
        std::map<std::vector<uint32_t>, bool> queued;
        for (size_t i = 0; i < N; i++) {
            if (in_world[i] || blas_cache.count(keys[i]) || queued.count(keys[i])) continue;
            uint32_t g = 0, nt = 0;
            for (uint32_t m : keys[i]) { bg.push_back(blas_geo(meshes[m], off, g++, 0u)); off += meshes[m]->index_count; nt += meshes[m]->index_count; }
            queued[keys[i]] = true;
            if (nt == 0) { blas_cache[keys[i]] = BlasInfo{ MAX_UINT, { 0, 0, 0, 0, 0, 0 }, 0u }; while (g--) bg.pop_back(); continue; }
            job_first.push_back(off); job_key.push_back(keys[i]);
        }
        if (!world_key.empty() && !blas_cache.count(world_key)) {
            for (size_t i = 0; i < N; i++) {
                if (!in_world[i]) continue;
                uint32_t g = 0;
                for (uint32_t m : keys[i]) { bg.push_back(blas_geo(meshes[m], off, g++, (uint32_t)i)); off += meshes[m]->index_count; }
            }
            job_first.push_back(off); job_key.push_back(world_key);
        }
        const size_t njobs = job_key.size();
        if (njobs) {
            std::vector<uint32_t> roots(njobs); std::vector<float> boxes(6 * njobs);
            if (!bvh_build_blas_batch(build_scratch, stream, bg, job_first, d_nodes.p, d_build_counters.p, (uint32_t)d_nodes.n, d_tris.p, d_tri_rot.p, d_tri_attrs.p, d_build_counters.p + 1, d_item_src.p,
                                      roots.data(), boxes.data())) { fail("BLAS build failed (details on stderr)"); return false; }
            for (size_t j = 0; j < njobs; j++) {
                BlasInfo info{}; info.root = roots[j]; info.tris = job_first[j + 1] - job_first[j];
                memcpy(info.box, &boxes[6 * j], 24);
                blas_cache[job_key[j]] = info;
            }
        }
    }
    { uint32_t c[2]; CHECK_HIP(this, hipMemcpy(c, d_build_counters.p, 8, hipMemcpyDeviceToHost)); blas_nodes_end = c[0]; blas_tris_end = c[1]; }

    lap("BLAS");
    // instance records + TLAS over the transformed visible instances and the world pseudo-instance (Accel.zig:394-484)
    std::vector<InstanceRec> irec(N + 1);
    std::vector<TlasInst> tinst; std::vector<TlasMesh> tmesh; std::vector<uint32_t> ids;
    // TLAS leaf boxes are computed on the GPU (k_instance_boxes) from the transformed vertices of every visible instance — up to 1.7x
    // tighter per axis than the transformed corners of the BLAS root box under rotation — on every TLAS build (S2: 5 M vertex transforms,
    // microseconds); MSNE_EXACT_INSTANCE_BOXES=0 selects the corner boxes.
    static const bool exact_boxes = [] { const char* e = getenv("MSNE_EXACT_INSTANCE_BOXES"); return !e || atoi(e) != 0; }();
    auto add_box = [&](const m34& T, const float box[6], uint32_t id, const std::vector<uint32_t>* mesh_ids) {
        TlasInst t{};
        memcpy(t.T, &T, 48); memcpy(t.blas_box, box, 24);
        t.mesh_begin = (uint32_t)tmesh.size();
        if (mesh_ids && exact_boxes) for (uint32_t mi : *mesh_ids) tmesh.push_back(TlasMesh{ meshes[mi]->positions.p, meshes[mi]->position_count, 0u });
        t.mesh_end = (uint32_t)tmesh.size(); t.exact = t.mesh_end > t.mesh_begin ? 1u : 0u;
        tinst.push_back(t); ids.push_back(id);
        coord_radius = std::max(coord_radius, coord_reach(T, box));
    };
    coord_radius = 0.0f;
    for (size_t i = 0; i < N; i++) {
        InstanceRec& r = irec[i];
        r.transform = instances[i].transform;
        r.world_to_instance = m34_inverse_affine(instances[i].transform);   // Accel.zig:430-432
        r.geo_offset = geo_offset[i]; r.pad = 0;
        r.flags = (instances[i].visible ? INST_FLAG_VISIBLE : 0u) | (is_identity(instances[i].transform) ? INST_FLAG_IDENTITY : 0u);
        r.blas_root = MAX_UINT;
        if (in_world[i]) continue;
        const BlasInfo& bi = blas_cache[keys[i]];
        r.blas_root = bi.root;
        if (!instances[i].visible || bi.root == MAX_UINT || !finite_transform(instances[i].transform)) continue;
        add_box(instances[i].transform, bi.box, (uint32_t)i, &keys[i]);
    }
    m34 ident; for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) ident.m[r][c] = r == c ? 1.0f : 0.0f;
    {
        InstanceRec& w = irec[N];
        w.transform = ident; w.world_to_instance = ident; w.geo_offset = 0; w.pad = 0; w.blas_root = MAX_UINT; w.flags = 0;
        if (!world_key.empty()) {
            const BlasInfo& bi = blas_cache[world_key];
            w.blas_root = bi.root; w.flags = INST_FLAG_VISIBLE | INST_FLAG_IDENTITY | INST_FLAG_WORLD;
            add_box(ident, bi.box, (uint32_t)N, nullptr);
        }
    }
    baked_origin_reach = origin_reach_wanted();
    bool padded = false;
    for (TlasInst& t : tinst) { m34 T; memcpy(&T, t.T, 48); t.cull_pad = instance_cull_pad(T, t.blas_box, baked_origin_reach); padded = padded || t.cull_pad != 0.0f; }
    if (!padded) baked_origin_reach = 3.0e38f;   // (identity instances only — S1: the traversal never changes space, nothing was grown for anything)
    lap("instance records");
    if (!d_instances.alloc(irec.size())) { fail("out of device memory (instances)"); return false; }
    CHECK_HIP(this, hipMemcpyAsync(d_instances.p, irec.data(), irec.size() * sizeof(InstanceRec), hipMemcpyHostToDevice, stream));
    root_in_blas = 0;
    if (ids.size() == 1 && ids[0] == (uint32_t)N) {
        tlas_root = irec[N].blas_root; root_in_blas = 1;     // nothing but static geometry: traversal starts inside the world BLAS
        const uint32_t zero = 0; CHECK_HIP(this, hipMemcpyAsync(d_build_counters.p + 2, &zero, 4, hipMemcpyHostToDevice, stream));
    } else if (!bvh_build_tlas(build_scratch, stream, tinst.data(), ids.data(), (uint32_t)ids.size(), tmesh.data(), (uint32_t)tmesh.size(), d_nodes.p, d_build_counters.p, (uint32_t)d_nodes.n, d_tlas_items.p, d_build_counters.p + 2, d_item_src.p, &tlas_root, d_inst_spheres.p)) { fail("TLAS build failed (details on stderr)"); return false; }

    // what an in-place update of this TLAS needs later: parent links of its nodes and leaf items, the leaf item of every instance, the records as uploaded
    h_irec = irec; built_in_world = in_world; item_of_instance.assign(N + 1, MAX_UINT);
    tlas_node_begin = tlas_node_end = blas_nodes_end; tlas_item_begin = 0;
    if (!root_in_blas && !ids.empty()) {
        uint32_t node_end = 0;
        CHECK_HIP(this, hipMemcpy(&node_end, d_build_counters.p, 4, hipMemcpyDeviceToHost));
        tlas_node_end = node_end;
        std::vector<uint32_t> h_items(ids.size());
        CHECK_HIP(this, hipMemcpy(h_items.data(), d_tlas_items.p, ids.size() * 4, hipMemcpyDeviceToHost));
        for (size_t it = 0; it < h_items.size(); it++) if (h_items[it] <= N) item_of_instance[h_items[it]] = (uint32_t)it;
        if (!d_tlas_node_parent.ensure(tlas_node_end - tlas_node_begin + 1) || !d_tlas_item_parent.ensure(ids.size() + 1)) { fail("out of device memory (TLAS links)"); return false; }
        bvh_tlas_links(stream, d_nodes.p, tlas_node_begin, tlas_node_end, tlas_item_begin, d_tlas_node_parent.p, d_tlas_item_parent.p, tlas_root);
        bvh_tlas_leaves(stream, d_tlas_items.p, d_instances.p, d_inst_spheres.p, (uint32_t)ids.size(), d_tlas_leaves.p);
        n_tlas_items = (uint32_t)ids.size();
    }
    if (timing) (void)hipStreamSynchronize(stream);
    lap("TLAS");
    // emissive-triangle alias table (Accel.zig:491-539): entry 0 = {count, sum of areas}
    std::vector<float> w; h_alias.assign(1, AliasEntry{ 0u, 0.0f, 0u, 0u, 0u });
    for (size_t i = 0; i < instances.size(); i++) for (size_t g = 0; g < instances[i].geos.size(); g++) {
        const GeometryRec& ge = instances[i].geos[g];
        if (!ge.sampled) continue;
        const MeshH* m = meshes[ge.mesh];
        for (uint32_t p = 0; p < m->index_count; p++) {
            const uint32_t* ix = &m->h_indices[3 * (size_t)p];
            const float* P = m->h_positions.data();
            const f3 p0 = m34_mul_point(instances[i].transform, F3(P[3 * (size_t)ix[0]], P[3 * (size_t)ix[0] + 1], P[3 * (size_t)ix[0] + 2]));
            const f3 p1 = m34_mul_point(instances[i].transform, F3(P[3 * (size_t)ix[1]], P[3 * (size_t)ix[1] + 1], P[3 * (size_t)ix[1] + 2]));
            const f3 p2 = m34_mul_point(instances[i].transform, F3(P[3 * (size_t)ix[2]], P[3 * (size_t)ix[2] + 1], P[3 * (size_t)ix[2] + 2]));
            w.push_back(length(cross(sub(p1, p0), sub(p2, p0))) / 2.0f);
            h_alias.push_back(AliasEntry{ 0u, 0.0f, (uint32_t)i, (uint32_t)g, p });
        }
    }
    float sum = 0.0f;
    if (!w.empty()) vose_alias(w, h_alias.data() + 1, sum);
    h_alias[0].alias = (uint32_t)w.size(); h_alias[0].select = sum;
    if (!d_alias.alloc(h_alias.size())) { fail("out of device memory (alias table)"); return false; }
    CHECK_HIP(this, hipMemcpyAsync(d_alias.p, h_alias.data(), h_alias.size() * sizeof(AliasEntry), hipMemcpyHostToDevice, stream));
    lights_dirty = true;
    CHECK_HIP(this, hipStreamSynchronize(stream));
    // ~200 B of scratch per primitive: a scene-sized scratch is given back, a TLAS-sized one (interactive instance edits, up to a million instances) is kept
    lap("alias table");
    if (bvh_scratch_capacity(build_scratch) > (1u << 20)) bvh_scratch_release(build_scratch);
    else if (bvh_scratch_arena_bytes(build_scratch) > (64u << 20)) bvh_scratch_release_arena(build_scratch);   // (the sweep's working set — ~120 B per position — of a BLAS build just under that size: S1 would keep 120 MB per context for its lifetime)
    lap("scratch release");
    if (timing) fprintf(stderr, "moonshine_amd rebuild (%zu instances):%s\n", N, t_report.c_str());
    accel_dirty = false; transform_edits.clear(); n_rebuilds++; refits_since_rebuild = 0;
    return true;
}

// Transform edits of instances that keep their place in the scene's structure (Accel.zig:567-601, hydra.zig:225-311): the instance records are
// overwritten, the instances' TLAS leaves get the boxes of their newly transformed vertices and the boxes on the way to the root are re-fitted in
// place — no rebuild (100 000 instances: 14 ms of rebuild against a fraction of a millisecond).
bool HdMoonshine::refit_tlas() {
    std::sort(transform_edits.begin(), transform_edits.end());
    transform_edits.erase(std::unique(transform_edits.begin(), transform_edits.end()), transform_edits.end());
    std::vector<TlasInst> tinst; std::vector<TlasMesh> tmesh; std::vector<uint32_t> items;
    static const bool exact_boxes = [] { const char* e = getenv("MSNE_EXACT_INSTANCE_BOXES"); return !e || atoi(e) != 0; }();
    for (uint32_t h : transform_edits) {
        if (h >= instances.size() || h >= h_irec.size() || item_of_instance[h] == MAX_UINT) return false;
        std::vector<uint32_t> key; for (auto& g : instances[h].geos) key.push_back(g.mesh);
        auto bi = blas_cache.find(key);
        if (bi == blas_cache.end() || bi->second.root == MAX_UINT) return false;
        InstanceRec& r = h_irec[h];
        r.transform = instances[h].transform; r.world_to_instance = m34_inverse_affine(instances[h].transform);
        r.flags = INST_FLAG_VISIBLE;   // (a refit never sees an identity transform or a hidden instance: those are rebuilds)
        if (transform_edits.size() <= 64) CHECK_HIP(this, hipMemcpyAsync(d_instances.p + h, &r, sizeof(InstanceRec), hipMemcpyHostToDevice, stream));
        TlasInst t{};
        memcpy(t.T, &instances[h].transform, 48); memcpy(t.blas_box, bi->second.box, 24);
        t.mesh_begin = (uint32_t)tmesh.size();
        if (exact_boxes) for (uint32_t mi : key) tmesh.push_back(TlasMesh{ meshes[mi]->positions.p, meshes[mi]->position_count, 0u });
        t.mesh_end = (uint32_t)tmesh.size(); t.exact = t.mesh_end > t.mesh_begin ? 1u : 0u;
        tinst.push_back(t); items.push_back(item_of_instance[h]);
        coord_radius = std::max(coord_radius, coord_reach(instances[h].transform, bi->second.box));
    }
    // the pads and spheres of the instances that are NOT edited were baked for origins up to baked_origin_reach (16 x the coord_radius of that build, or more): an edit
    // that carries an instance beyond it — so that rays starting ON it lie outside what the others were grown for — is a rebuild (advisor, round 5)
    if (coord_radius * 1.01f > baked_origin_reach) return false;
    for (TlasInst& t : tinst) { m34 T; memcpy(&T, t.T, 48); t.cull_pad = instance_cull_pad(T, t.blas_box, baked_origin_reach); }
    if (transform_edits.size() > 64) CHECK_HIP(this, hipMemcpyAsync(d_instances.p, h_irec.data(), h_irec.size() * sizeof(InstanceRec), hipMemcpyHostToDevice, stream));   // many edits: the whole table in one copy
    if (!build_scratch && !(build_scratch = bvh_scratch_create())) { fail("out of host memory"); return false; }
    if (!bvh_refit_tlas(build_scratch, stream, tinst.data(), tmesh.data(), (uint32_t)tmesh.size(), items.data(), (uint32_t)items.size(), d_nodes.p, tlas_node_begin, tlas_node_end - tlas_node_begin, tlas_item_begin, n_tlas_items,
                        d_tlas_node_parent.p, d_tlas_item_parent.p, transform_edits.data(), d_inst_spheres.p)) return false;
    bvh_tlas_leaves(stream, d_tlas_items.p, d_instances.p, d_inst_spheres.p, n_tlas_items, d_tlas_leaves.p);   // (the edited instances' matrices; all of them rewritten: microseconds)
    transform_edits.clear(); n_tlas_updates++; refits_since_rebuild++;
    return true;
}

bool HdMoonshine::ensure_scene() {
    if (textures_dirty && !upload_textures()) return false;
    if ((materials_dirty || !material_updates.empty()) && !upload_materials()) return false;
    if (blas_indexed != (opts.indexed_attributes != 0)) accel_dirty = true;
    // Transform edits re-fit the TLAS in place (every edit its own thread: bvh_refit_tlas) while they are a minority of the instances; when a quarter of the scene moves
    // at once the tree's shape is stale anyway and the rebuild is cheap (17 ms for 100 000 instances).  A re-fitted tree keeps its shape and its re-made grids round
    // outward, so its boxes only ever grow: after 64 re-fits in a row it is rebuilt (the reference's UPDATE-mode builds degrade the same way, Accel.zig:567-601).
    if (!accel_dirty && origin_needed > baked_origin_reach) accel_dirty = true;   // a primary ray from farther out than the instances' culling volumes were grown for
    if (!accel_dirty && !transform_edits.empty() &&
        (transform_edits.size() > std::max<size_t>(256, instances.size() / 4) || refits_since_rebuild >= 64 || !refit_tlas())) accel_dirty = true;
    if (accel_dirty) geometry_edits.clear();   // (the rebuild writes the whole geometry table from the instances)
    if (accel_dirty && !rebuild_accel()) return false;
    // Accel.recordUpdateSingleMaterial (Accel.zig:609-628): one 4-byte update of the geometry's record — the acceleration structure, the triangle records and the
    // alias table (areas only) do not know materials; the gathered light triangles do (material index + emissive descriptor) and are gathered again
    for (auto& e : geometry_edits) {
        const GeometryRec& g = instances[e.first].geos[e.second];
        CHECK_HIP(this, hipMemcpyAsync(reinterpret_cast<char*>(d_geometries.p + h_irec[e.first].geo_offset + e.second) + offsetof(GeometryRec, material), &g.material, 4, hipMemcpyHostToDevice, stream));
        if (g.sampled) lights_dirty = true;
    }
    if (!geometry_edits.empty()) { CHECK_HIP(this, hipStreamSynchronize(stream)); geometry_edits.clear(); }
    if (lights_dirty || lights_indexed != opts.indexed_attributes) {
        const uint32_t count = h_alias.empty() ? 0u : h_alias[0].alias;
        if (count) {
            if (!d_light_tris.alloc((size_t)count + 1)) { fail("out of device memory (light triangles)"); return false; }
            launch_light_tris(stream, scene_view(), opts.indexed_attributes, (uint32_t)instances.size(), d_light_tris.p);
            CHECK_HIP(this, hipStreamSynchronize(stream));
        }
        lights_dirty = false; lights_indexed = opts.indexed_attributes;
    }
    return true;
}

SceneView HdMoonshine::scene_view() const {
    SceneView v{};
    v.nodes = d_nodes.p; v.tris = d_tris.p; v.tri_rot = d_tri_rot.p; v.tri_attrs = d_tri_attrs.p; v.tlas_items = d_tlas_items.p; v.tlas_leaves = d_tlas_leaves.p; v.instances = d_instances.p; v.geometries = d_geometries.p;
    v.meshes = d_meshes.p; v.materials = d_materials.p; v.textures = d_texdesc.p; v.texels = d_texels.p; v.srgb_lut = d_srgb.p; v.alias = d_alias.p;
    if (!h_alias.empty()) { v.alias_count = h_alias[0].alias; v.alias_sum = h_alias[0].select; }
    v.light_tris = d_light_tris.p;
    v.env = env; v.tlas_root = tlas_root; v.root_in_blas = root_in_blas;
    v.coord_slack = 1.5e-6f * coord_radius;
    return v;
}

// ---------------- background (BackgroundManager.zig:142-394) ----------------
bool HdMoonshine::set_background(const float* rgba, Extent2D e) {
    if (!rgba || e.width == 0 || e.height == 0) { fail("background: bad arguments"); return false; }
    uint32_t S = 1; while (S * 2 <= e.height && S * 2 != 0) S *= 2;      // floorPowerOfTwo(height), BackgroundManager.zig:154
    if (S > 1024) S = 1024;                                               // maximum_equal_area_map_size :132
    uint32_t mips = 1; while ((S >> (mips - 1)) > 1) mips++;
    DevBuf<float4> src;
    if (!src.alloc((size_t)e.width * e.height) || !d_env_rgb.alloc((size_t)S * S)) { fail("out of device memory (background)"); return false; }
    size_t total = 0; uint32_t off[12] = { 0 };
    for (uint32_t l = 0; l < mips; l++) { off[l] = (uint32_t)total; total += (size_t)(S >> l) * (S >> l); }
    if (!d_env_lum.alloc(total)) { fail("out of device memory (background)"); return false; }
    CHECK_HIP(this, hipMemcpyAsync(src.p, rgba, (size_t)e.width * e.height * 16, hipMemcpyHostToDevice, stream));
    launch_env_build(stream, src.p, e.width, e.height, d_env_rgb.p, d_env_lum.p, off, S, mips);
    size_t nquads = 0; uint32_t qoff[12] = { 0 };
    for (uint32_t l = 0; l + 1 < mips; l++) { qoff[l] = (uint32_t)nquads; nquads += (size_t)((S >> l) / 2) * ((S >> l) / 2); }
    if (nquads) {
        if (!d_env_quads.alloc(nquads)) { fail("out of device memory (background)"); return false; }
        launch_env_quads(stream, d_env_lum.p, off, d_env_quads.p, qoff, S, mips);
    }
    float top = 0.0f;
    CHECK_HIP(this, hipMemcpyAsync(&top, d_env_lum.p + off[mips - 1], 4, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(this, hipStreamSynchronize(stream));
    env.rgb = d_env_rgb.p; env.lum = d_env_lum.p; env.size = S; env.mip_count = mips;
    env.quads = nquads ? d_env_quads.p : nullptr; env.top = top;
    for (int l = 0; l < 12; l++) { env.lum_offset[l] = off[l]; env.quad_offset[l] = qoff[l]; }
    clear_all_sensors();
    return true;
}

// ---------------- wavefront buffers ----------------
bool HdMoonshine::ensure_wavefront(size_t npaths, size_t slots, int npipes) {
    for (int k = 0; k < npipes; k++) {
        Pipe& pp = pipes[k];
        if (!pp.s0 && (hipStreamCreateWithFlags(&pp.s0, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&pp.s1, hipStreamNonBlocking) != hipSuccess)) { fail("cannot create HIP streams"); return false; }
        if (((npaths + 255) & ~(size_t)255) + (size_t)QUEUE_SUBS * 256 > pp.cap) {
            const size_t c = ((npaths + 255) & ~(size_t)255) + (size_t)QUEUE_SUBS * 256;   // (a queue's extent rounds up to a whole round of sub-queue tiles: msne_device.h)
            pp.cap = 0;
            if (!pp.paths[0].ensure(c) || !pp.paths[1].ensure(c) || !pp.hit_u.alloc(4 * c)) { fail("out of device memory (wavefront state)"); return false; }
            pp.cap = c; pp.shq_samples = 0;
        }
        const uint32_t ns = std::max(1u, opts.env_samples + opts.mesh_samples);   // shadow-queue entries per path
        if ((unsigned long long)ns * pp.cap > (1ull << SHQ_POS_BITS)) {   // (only a film of more pixels than inflight_budget() gets here: one sample of it is the smallest batch)
            fail("film too large for the light-sample queue: " + std::to_string(pp.cap) + " paths x " + std::to_string(ns) + " light samples per bounce exceed 2^" + std::to_string(SHQ_POS_BITS) + " entries");
            return false;
        }
        if (ns > pp.shq_samples) {
            if (!pp.shq_f.alloc(4 * 4 * (size_t)ns * pp.cap)) { pp.shq_samples = 0; fail("out of device memory (shadow queue)"); return false; }
            pp.shq_samples = ns;
        }
        const size_t nc = (size_t)opts.max_bounces + 5;   // bounces 0 .. max_bounces + 2, the entry k_shade of the last one appends to, and the probe's
        if (pp.counters.n < nc && !pp.counters.alloc(nc)) return false;
        if (!pp.totals.p) { if (!pp.totals.alloc(1)) return false; if (hipMemsetAsync(pp.totals.p, 0, sizeof(Totals), stream) != hipSuccess) return false; }
        if (!pp.spill.p && (!pp.spill.alloc(trace_spill_words(trace_grid)) || !pp.spill2.alloc(trace_spill_words(trace_grid)))) { fail("out of device memory (traversal spill)"); return false; }
    }
    if (slots > lbuf_cap) { if (!d_lbuf.alloc(slots)) { fail("out of device memory (sample buffer)"); lbuf_cap = 0; return false; } lbuf_cap = slots; }
    if (!d_overflow.p) {
        if (!d_overflow.alloc(1) || !d_trace_stats.alloc(44)) { fail("out of device memory"); return false; }
        if (hipMemsetAsync(d_overflow.p, 0, 4, stream) != hipSuccess || hipMemsetAsync(d_trace_stats.p, 0, 352, stream) != hipSuccess) return false;
    }
    return hipStreamSynchronize(stream) == hipSuccess;
}

static CameraConsts make_camera(const Lens& lens, uint32_t W, uint32_t H) {   // camera.hlsl:14-29, evaluated once per launch instead of per thread
    CameraConsts c;
    const f3 origin = F3(lens.origin.x, lens.origin.y, lens.origin.z), forward = F3(lens.forward.x, lens.forward.y, lens.forward.z), up = F3(lens.up.x, lens.up.y, lens.up.z);
    const float aspect = (float)W / (float)H;
    const f3 w = scale(forward, -1.0f);
    const f3 u = normalize(cross(up, w));
    const f3 v = cross(w, u);
    const float h = det_tanf(lens.vfov / 2.0f);
    const float viewport_height = 2.0f * h * lens.focus_distance;
    const float viewport_width = aspect * viewport_height;
    c.origin = origin; c.u = u; c.v = v;
    c.horizontal = scale(u, viewport_width);
    c.vertical = scale(v, viewport_height);
    c.llc = sub(sub(sub(origin, divs(c.horizontal, 2.0f)), divs(c.vertical, 2.0f)), scale(w, lens.focus_distance));
    c.aperture = lens.aperture;
    return c;
}

bool HdMoonshine::readback(SensorH* s) {
    launch_unpack_film(stream, 1024, s->shard, s->film_packed.p, 1, s->shard.shard_index, 0, s->film_full.p);
    CHECK_HIP(this, hipMemcpyAsync(s->host, s->film_full.p, (size_t)s->extent.width * s->extent.height * 16, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(this, hipStreamSynchronize(stream));
    return true;
}

bool HdMoonshine::render(uint32_t sensor, uint32_t lens, uint32_t launches, bool do_readback) {
    if (sensor >= sensors.size() || lens >= lenses.size()) { fail("render: bad sensor or lens handle"); return false; }
    origin_needed = 0.0f; need_origin(lenses[lens].origin.x, lenses[lens].origin.y, lenses[lens].origin.z, lenses[lens].aperture);   // camera rays start on the lens (camera.hlsl:31-40)
    if (!ensure_scene()) return false;
    // normals / texcoords are read by vertex index (indexed_attributes, the glTF path) or by corner = 3 * triangle + k (Hydra's
    // face-varying path, world.hlsl:127-135): the array every referenced mesh handed over must cover what this pipeline reads
    for (const InstanceH& in : instances) for (const GeometryRec& g : in.geos) {
        const MeshH* m = meshes[g.mesh];
        if (!m->has_normals && !m->has_texcoords) continue;
        const uint64_t need = opts.indexed_attributes ? (uint64_t)m->max_index + 1u : 3ull * m->index_count;
        if (m->attribute_count < need) {
            fail("mesh " + std::to_string(g.mesh) + " has " + std::to_string(m->attribute_count) + " normals/texcoords, the pipeline (indexed_attributes = "
                 + (opts.indexed_attributes ? "true" : "false") + ") reads " + std::to_string(need));
            return false;
        }
    }
    SensorH* s = sensors[sensor];
    const size_t P = s->shard.pixels;
    const uint32_t spr = opts.samples_per_run;
    if (P == 0 || spr == 0) { if (do_readback) return readback(s); return true; }
    // how many samples are traced concurrently: whole launches are batched (nb launches in flight, folded into the
    // film in launch order by k_film — bit-identical to nb sequential dispatches, because the RNG is keyed by the sample
    // index, main.hlsl:85); a launch too large for the in-flight budget is split into chunks of its samples instead.
    const size_t per_launch = P * (size_t)spr;
    const size_t max_inflight = inflight_budget();   // (shadows the member: the shadow queue's 32-bit positions bound a batch too)
    const uint32_t max_batch = per_launch <= max_inflight ? (uint32_t)std::max<size_t>(1, max_inflight / per_launch) : 1u;
    const uint32_t chunk = per_launch <= max_inflight ? spr : (uint32_t)std::max<size_t>(1, max_inflight / P);
    const uint32_t first_batch = chunk == spr ? std::min<uint32_t>(max_batch, launches) : 1u;
    // pipes for a batch of nb launches: one when the batch is huge or cannot be split, else up to n_pipes
    auto pipes_for = [&](uint32_t nb) -> int { return (chunk != spr || nb < 2 || per_launch * nb >= single_pipe_paths) ? 1 : (int)std::min<uint32_t>((uint32_t)n_pipes, nb); };
    {
        const int K = pipes_for(first_batch);
        const size_t per_pipe = chunk == spr ? per_launch * ((first_batch + K - 1) / K) : P * (size_t)chunk;
        if (!ensure_wavefront(per_pipe, chunk == spr ? per_launch * first_batch : P * (size_t)chunk, K)) return false;
    }
    const SceneView sv = scene_view();
    const CameraConsts cam = make_camera(lenses[lens], s->extent.width, s->extent.height);
    // every texture 1x1 (constant material parameters: the glTF factors of World.zig:44-228, every synthetic scene): k_shade without the bilinear sampler
    static const int force_tex = [] { const char* e = getenv("MSNE_SHADE_TEXTURED"); return e ? atoi(e) : -1; }();
    bool textured = false;
    for (const auto& t : textures) if (t.w != 1 || t.h != 1) { textured = true; break; }
    if (force_tex >= 0) textured = textured || force_tex != 0;
    events_used = 0; spans.clear();
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    ev_begin = next_event(); ev_end = next_event();
    if (ev_begin) (void)hipEventRecord(ev_begin, stream);
    // hits b = 0 .. max_bounces + 1.  The last pass creates nothing (integrator.hlsl:128 ends every path before its light samples and its next direction): it adds emission,
    // the light samples of the pass before and retires that pass's zombies, and needs no shadow rays after it.
    const uint32_t max_iter = opts.max_bounces + 2;
    auto timed2 = [&](int kind, hipStream_t st_, auto&& fn) {
        if (!profile) { fn(); return; }
        hipEvent_t a = next_event(), b = next_event();
        const size_t ia = events_used - 2, ib = events_used - 1;
        if (a) (void)hipEventRecord(a, st_);
        fn();
        if (b) (void)hipEventRecord(b, st_);
        spans.push_back(Span{ ia, ib, kind });
    };
    // One wavefront pass of a pipe over `ns` samples per pixel starting at sample index `first_sample`; finished samples
    // go to lbuf[0 .. ns*P).  Two streams per pipe: k_trace_closest / k_shade / bookkeeping on s0, k_trace_shadow on s1 —
    // the shadow rays of bounce b are only needed by k_shade(b+1), so k_trace_shadow(b) overlaps k_trace_closest(b+1).
    auto trace_pass = [&](Pipe& pp, uint32_t first_sample, uint32_t ns, float4* lbuf) -> bool {
        const size_t qc = (size_t)pp.shq_samples * pp.cap;   // shadow-queue capacity: o, d, then the two contribution buffers (by bounce parity)
        const HitBuf hits{ reinterpret_cast<uint4*>(pp.hit_u.p) };
        float4* const shq_f4 = reinterpret_cast<float4*>(pp.shq_f.p);
        float4* const contrib[2] = { shq_f4 + 2 * qc, shq_f4 + 3 * qc };
        const PathState st[2] = { pp.paths[0].view(), pp.paths[1].view() };
        BounceCounters* cnt = pp.counters.p;   // [b] = the queues of bounce b: nothing to rotate or reset between kernels
        CHECK_HIP(this, hipMemsetAsync(cnt, 0, ((size_t)max_iter + 2) * sizeof(BounceCounters), pp.s0));
        launch_raygen(pp.s0, shade_grid, s->shard, cam, opts, first_sample, ns, st[0], cnt);
        hipEvent_t shadow_done = nullptr;
        const size_t batch_paths = (size_t)ns * P;
        for (uint32_t b = 0; b < max_iter; b++) {
            const bool in_order = serial_mode == 1 || (serial_mode < 0 && b < serial_bounces && batch_paths >= serial_min_paths);
            const hipStream_t sh_stream = in_order ? pp.s0 : pp.s1;
            const PathState& cur = st[b & 1]; const PathState& nxt = st[(b + 1) & 1];
            timed2(0, pp.s0, [&] { launch_trace_closest(pp.s0, trace_grid, trace_stats, sv, cur, hits, cnt + b, pp.spill.p, d_overflow.p, d_trace_stats.p, refill); });
            if (shadow_done) CHECK_HIP(this, hipStreamWaitEvent(pp.s0, shadow_done, 0));   // k_shade(b) consumes the results of k_trace_shadow(b-1)
            const ShadowQueue shq{ shq_f4, shq_f4 + qc, contrib[b & 1] };
            timed2(2, pp.s0, [&] { launch_shade(pp.s0, shade_k_grid, sv, opts, cur, hits, nxt, shq, contrib[(b + 1) & 1], lbuf, cnt + b, b == 0, textured); });
            hipEvent_t shade_done = next_event();
            if (!shade_done) { fail("hipEventCreate failed"); return false; }
            CHECK_HIP(this, hipEventRecord(shade_done, pp.s0));
            if (b + 1 == max_iter) { shadow_done = nullptr; break; }   // (k_shade of the last pass wrote no shadow rays)
            CHECK_HIP(this, hipStreamWaitEvent(sh_stream, shade_done, 0));
            timed2(1, sh_stream, [&] { launch_trace_shadow(sh_stream, trace_grid, trace_stats, sv, shq, cnt + b + 1, pp.spill2.p, d_overflow.p, d_trace_stats.p, std::max(1u, refill * 3 / 4)); });   // the shadow queue has unused entries (light samples with pdf 0): refill a little sooner (12 idle lanes: 25.4 ms against 26.0 at 8 and 25.3 at 16, S1)
            shadow_done = next_event();
            if (!shadow_done) { fail("hipEventCreate failed"); return false; }
            CHECK_HIP(this, hipEventRecord(shadow_done, sh_stream));
            if (b >= 15 && (b & 3) == 3) {   // long tails (max_bounces = 1024 offline): poll the queue length every 4 bounces
                BounceCounters next;
                CHECK_HIP(this, hipMemcpyAsync(&next, &cnt[b + 1], sizeof next, hipMemcpyDeviceToHost, pp.s0));
                CHECK_HIP(this, hipStreamSynchronize(pp.s0));
                uint32_t n_next = 0;
                for (uint32_t k = 0; k < QUEUE_SUBS; k++) n_next += next.sub[k].n_paths;
                if (n_next == 0) break;
            }
        }
        if (shadow_done) CHECK_HIP(this, hipStreamWaitEvent(pp.s0, shadow_done, 0));
        launch_account(pp.s0, cnt, max_iter, pp.totals.p);
        return true;
    };
    // fork the pipes after everything queued on the main stream so far, join them back before k_film
    auto run_batch = [&](uint32_t first_sample, uint32_t ns_total, uint32_t ns_unit, int K) -> bool {
        hipEvent_t fork = next_event();
        if (!fork) { fail("hipEventCreate failed"); return false; }
        CHECK_HIP(this, hipEventRecord(fork, stream));
        const uint32_t units = ns_total / ns_unit;
        uint32_t u0 = 0;
        for (int k = 0; k < K; k++) {
            const uint32_t nu = units / K + ((uint32_t)k < units % K ? 1u : 0u);
            if (!nu) continue;
            Pipe& pp = pipes[k];
            CHECK_HIP(this, hipStreamWaitEvent(pp.s0, fork, 0));
            if (!trace_pass(pp, first_sample + u0 * ns_unit, nu * ns_unit, d_lbuf.p + (size_t)u0 * ns_unit * P)) return false;
            hipEvent_t join = next_event();
            if (!join) { fail("hipEventCreate failed"); return false; }
            CHECK_HIP(this, hipEventRecord(join, pp.s0));
            CHECK_HIP(this, hipStreamWaitEvent(stream, join, 0));
            u0 += nu;
        }
        return true;
    };
    for (uint32_t l = 0; l < launches;) {
        if (chunk == spr) {
            const uint32_t nb = std::min(max_batch, launches - l);
            if (!run_batch(s->sample_count, nb * spr, spr, pipes_for(nb))) return false;
            launch_film(stream, shade_grid, s->shard, opts, d_lbuf.p, spr, nb, 1, 1, s->sample_count, s->color.p, s->film_packed.p);
            s->sample_count += nb * spr;   // hydra.zig:360, nb times
            stats.launches += nb; l += nb;
        } else {
            for (uint32_t s0 = 0; s0 < spr; s0 += chunk) {
                const uint32_t sc = std::min(chunk, spr - s0);
                if (!run_batch(s->sample_count + s0, sc, sc, 1)) return false;
                launch_film(stream, shade_grid, s->shard, opts, d_lbuf.p, sc, 1, s0 == 0, s0 + sc == spr, s->sample_count, s->color.p, s->film_packed.p);
            }
            s->sample_count += spr;
            stats.launches++; l++;
        }
    }
    if (ev_end) (void)hipEventRecord(ev_end, stream);
    if (do_readback) { if (!readback(s)) return false; }
    else CHECK_HIP(this, hipStreamSynchronize(stream));
    if (!check_overflow()) { s->sample_count = 0; return false; }   // the film holds truncated traversals: the next render starts it over (Sensor.clear)
    float ms = 0.0f;
    if (ev_begin && ev_end && hipEventElapsedTime(&ms, ev_begin, ev_end) == hipSuccess) stats.render_ms += ms;
    launch_times.clear();
    for (const Span& sp : spans) {
        if (hipEventElapsedTime(&ms, events[sp.a], events[sp.b]) != hipSuccess) continue;
        launch_times.emplace_back(sp.kind, ms);
        if (sp.kind == 0) { stats.trace_closest_ms += ms; stats.trace_closest_launches++; }
        else if (sp.kind == 1) { stats.trace_shadow_ms += ms; stats.trace_shadow_launches++; }
        else { stats.shade_ms += ms; stats.shade_launches++; }
    }
    return true;
}

HdMoonshine::~HdMoonshine() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    for (auto& pp : pipes) { if (pp.s0) { (void)hipStreamSynchronize(pp.s0); (void)hipStreamDestroy(pp.s0); } if (pp.s1) { (void)hipStreamSynchronize(pp.s1); (void)hipStreamDestroy(pp.s1); } }
    for (auto* m : meshes) delete m;
    for (auto* s : sensors) { if (s->host) (void)hipHostFree(s->host); delete s; }
    for (auto e : events) (void)hipEventDestroy(e);
    bvh_scratch_destroy(build_scratch);
    if (stream) (void)hipStreamDestroy(stream);
}

// =====================================================================================
//                                       C ABI
// =====================================================================================
#define LOCK(c) std::lock_guard<std::mutex> lock_((c)->mutex)

extern "C" {

HdMoonshine* MsneCreate(const MsneConfig* cfg_in) {
    MsneConfig cfg{ -1, 0, 0, 0 };
    if (cfg_in) cfg = *cfg_in;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_error = "no HIP device available: libmoonshine_amd requires an MI355X (gfx950) GPU"; return nullptr; }
    int dev = cfg.device;
    if (dev < 0) { const char* e = getenv("MSNE_DEVICE"); dev = e ? atoi(e) : 0; }
    if (dev >= ndev) { g_create_error = "requested HIP device does not exist"; return nullptr; }
    if (cfg.tile_size == 0) cfg.tile_size = MSNE_DEFAULT_TILE_SIZE;
    if (cfg.shard_count == 0) cfg.shard_count = 1;
    if (cfg.shard_index >= cfg.shard_count) { g_create_error = "shard_index >= shard_count"; return nullptr; }
    HdMoonshine* c = new (std::nothrow) HdMoonshine();
    if (!c) { g_create_error = "out of host memory"; return nullptr; }
    c->device = dev; c->cfg = cfg;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { g_create_error = "cannot create HIP stream"; delete c; return nullptr; }
    hipDeviceProp_t prop{};
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) { g_create_error = "hipGetDeviceProperties failed"; delete c; return nullptr; }
    { c->trace_grid = prop.multiProcessorCount * trace_blocks_per_cu(); c->shade_grid = prop.multiProcessorCount * 8; c->shade_k_grid = prop.multiProcessorCount * 96; }   // k_shade: workgroups differ in cost (what their 256 paths hit); 96 per CU instead of 8 evens the CUs out (-6 %)
    if (const char* e = getenv("MSNE_SHADE_BLOCKS_PER_CU")) c->shade_k_grid = prop.multiProcessorCount * std::max(1, atoi(e));
    if (const char* e = getenv("MSNE_MAX_INFLIGHT")) c->max_inflight = (size_t)atoll(e);
    if (const char* e = getenv("MSNE_SERIAL")) c->serial_mode = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("MSNE_PIPES")) c->n_pipes = std::max(1, std::min((int)HdMoonshine::MAX_PIPES, atoi(e)));
    if (const char* e = getenv("MSNE_SINGLE_PIPE_PATHS")) c->single_pipe_paths = (size_t)atoll(e);
    if (const char* e = getenv("MSNE_REFILL")) c->refill = (uint32_t)std::max(1, std::min(64, atoi(e)));
    if (const char* e = getenv("MSNE_TRACE_BLOCKS_PER_CU")) c->trace_grid = prop.multiProcessorCount * atoi(e);
    c->opts = PipelineOpts{ 1, 1024, 0, 0, 0, 0, 0 };                 // hydra.zig:97-105
    const float white[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    if (!c->set_background(white, Extent2D{ 1, 1 })) { g_create_error = c->last_error; delete c; return nullptr; }   // addDefaultBackground
    return c;
}
HdMoonshine* HdMoonshineCreate(void) { return MsneCreate(nullptr); }
void HdMoonshineDestroy(HdMoonshine* c) { if (!c) return; { LOCK(c); (void)c->bind(); } delete c; }

const char* MsneGetLastError(const HdMoonshine* c) { return c ? c->last_error.c_str() : g_create_error.c_str(); }

int64_t MsneCreateMesh(HdMoonshine* c, const F32x3* positions, const F32x3* normals, const F32x2* texcoords, size_t position_count, size_t attribute_count, const U32x3* indices, size_t index_count) {
    LOCK(c);
    if (!c->bind()) return -1;
    if (!positions || !indices || position_count == 0 || index_count == 0) { c->fail("mesh: positions and indices are required"); return -1; }
    if ((normals || texcoords) && attribute_count == 0) { c->fail("mesh: attribute_count is zero"); return -1; }
    uint32_t max_index = 0;
    for (size_t i = 0; i < index_count; i++) {
        if (indices[i].x >= position_count || indices[i].y >= position_count || indices[i].z >= position_count) { c->fail("mesh: index out of range"); return -1; }
        max_index = std::max(max_index, std::max(indices[i].x, std::max(indices[i].y, indices[i].z)));
    }
    MeshH* m = new MeshH();
    m->max_index = max_index;
    m->position_count = (uint32_t)position_count; m->attribute_count = (uint32_t)attribute_count; m->index_count = (uint32_t)index_count;
    m->has_normals = normals != nullptr; m->has_texcoords = texcoords != nullptr;
    bool ok = m->positions.alloc(3 * position_count) && m->indices.alloc(3 * index_count);
    if (ok && normals) ok = m->normals.alloc(3 * attribute_count);
    if (ok && texcoords) ok = m->texcoords.alloc(2 * attribute_count);
    if (!ok) { delete m; c->fail("out of device memory (mesh)"); return -1; }
    m->h_positions.assign((const float*)positions, (const float*)positions + 3 * position_count);
    m->h_indices.assign((const uint32_t*)indices, (const uint32_t*)indices + 3 * index_count);
    bool up = hipMemcpyAsync(m->positions.p, positions, 12 * position_count, hipMemcpyHostToDevice, c->stream) == hipSuccess
           && hipMemcpyAsync(m->indices.p, indices, 12 * index_count, hipMemcpyHostToDevice, c->stream) == hipSuccess;
    if (up && normals) up = hipMemcpyAsync(m->normals.p, normals, 12 * attribute_count, hipMemcpyHostToDevice, c->stream) == hipSuccess;
    if (up && texcoords) up = hipMemcpyAsync(m->texcoords.p, texcoords, 8 * attribute_count, hipMemcpyHostToDevice, c->stream) == hipSuccess;
    if (up) up = hipStreamSynchronize(c->stream) == hipSuccess;   // MeshManager.upload copies: the caller keeps ownership
    if (!up) { delete m; c->fail("mesh upload failed"); return -1; }
    c->meshes.push_back(m);
    return (int64_t)c->meshes.size() - 1;
}
MeshHandle HdMoonshineCreateMesh(HdMoonshine* c, const F32x3* positions, const F32x3* normals, const F32x2* texcoords, size_t position_count, const U32x3* indices, size_t index_count) {
    const int64_t h = MsneCreateMesh(c, positions, normals, texcoords, position_count, index_count * 3, indices, index_count);   // hydra.zig:379-380
    return h < 0 ? 0xFFFFFFFFu : (MeshHandle)h;
}

int64_t MsneCreateTexture(HdMoonshine* c, const void* bytes, Extent2D e, MsneTextureFormat f) { LOCK(c); return add_texture(c, bytes, e.width, e.height, (int)f); }
ImageHandle HdMoonshineCreateSolidTexture1(HdMoonshine* c, float v, const char*) { LOCK(c); return (ImageHandle)add_texture(c, &v, 1, 1, MSNE_FORMAT_R32_SFLOAT); }
ImageHandle HdMoonshineCreateSolidTexture2(HdMoonshine* c, F32x2 v, const char*) { LOCK(c); return (ImageHandle)add_texture(c, &v, 1, 1, MSNE_FORMAT_R32G32_SFLOAT); }
ImageHandle HdMoonshineCreateSolidTexture3(HdMoonshine* c, F32x3 v, const char*) { LOCK(c); const float f[4] = { v.x, v.y, v.z, 0.0f }; return (ImageHandle)add_texture(c, f, 1, 1, MSNE_FORMAT_R32G32B32A32_SFLOAT); }   // stored as f32x4, MaterialManager.zig:380-388
ImageHandle HdMoonshineCreateRawTexture(HdMoonshine* c, uint8_t* data, Extent2D e, TextureFormat f, const char*) {
    LOCK(c); return (ImageHandle)add_texture(c, data, e.width, e.height, f == f16x4 ? MSNE_FORMAT_R16G16B16A16_SFLOAT : MSNE_FORMAT_R8G8B8A8_SRGB);   // hydra.zig:47-52
}

int64_t MsneCreateMaterial(HdMoonshine* c, const MsneMaterialDesc* d) {
    LOCK(c);
    if (!d || d->type > MSNE_MATERIAL_STANDARD_PBR) { c->fail("material: bad descriptor"); return -1; }
    const size_t nt = c->textures.size();
    if (d->normal >= nt || d->emissive >= nt) { c->fail("material: unknown texture handle"); return -1; }
    if ((d->type == MSNE_MATERIAL_LAMBERT || d->type == MSNE_MATERIAL_STANDARD_PBR) && d->color >= nt) { c->fail("material: unknown color texture"); return -1; }
    if (d->type == MSNE_MATERIAL_STANDARD_PBR && (d->metalness >= nt || d->roughness >= nt)) { c->fail("material: unknown texture handle"); return -1; }
    c->materials.push_back(MaterialRec{ d->normal, d->emissive, d->type, d->color, d->metalness, d->roughness, d->ior, 0u });
    c->materials_dirty = true;
    return (int64_t)c->materials.size() - 1;
}
MaterialHandle HdMoonshineCreateMaterial(HdMoonshine* c, Material m) {
    MsneMaterialDesc d{ m.normal, m.emissive, MSNE_MATERIAL_STANDARD_PBR, m.color, m.metalness, m.roughness, m.ior };
    const int64_t h = MsneCreateMaterial(c, &d);
    return h < 0 ? 0xFFFFFFFFu : (MaterialHandle)h;
}
static void defer_material(HdMoonshine* c, MaterialHandle m, int field, ImageHandle t, float ior) {
    LOCK(c); MaterialUpdate& u = c->material_updates[m]; u.has[field] = true; if (field < 5) u.tex[field] = t; else u.ior = ior;
}
void HdMoonshineSetMaterialNormal(HdMoonshine* c, MaterialHandle m, ImageHandle t) { defer_material(c, m, 0, t, 0.0f); }
void HdMoonshineSetMaterialEmissive(HdMoonshine* c, MaterialHandle m, ImageHandle t) { defer_material(c, m, 1, t, 0.0f); }
void HdMoonshineSetMaterialColor(HdMoonshine* c, MaterialHandle m, ImageHandle t) { defer_material(c, m, 2, t, 0.0f); }
void HdMoonshineSetMaterialMetalness(HdMoonshine* c, MaterialHandle m, ImageHandle t) { defer_material(c, m, 3, t, 0.0f); }
void HdMoonshineSetMaterialRoughness(HdMoonshine* c, MaterialHandle m, ImageHandle t) { defer_material(c, m, 4, t, 0.0f); }
void HdMoonshineSetMaterialIOR(HdMoonshine* c, MaterialHandle m, float ior) { defer_material(c, m, 5, 0, ior); }

InstanceHandle HdMoonshineCreateInstance(HdMoonshine* c, Mat3x4 t, const Geometry* geos, size_t n, bool visible) {
    LOCK(c);
    InstanceH in; memcpy(&in.transform, &t, sizeof(m34)); in.visible = visible;
    for (size_t i = 0; i < n; i++) in.geos.push_back(GeometryRec{ geos[i].mesh, geos[i].material, geos[i].sampled ? 1u : 0u });
    c->instances.push_back(std::move(in));
    c->accel_dirty = true; c->clear_all_sensors();           // hydra.zig:483-493
    return (InstanceHandle)c->instances.size() - 1;
}
void HdMoonshineSetInstanceVisibility(HdMoonshine* c, InstanceHandle h, bool v) { LOCK(c); if (h >= c->instances.size()) return; c->instances[h].visible = v; c->accel_dirty = true; c->clear_all_sensors(); }
void HdMoonshineDestroyInstance(HdMoonshine* c, InstanceHandle h) { HdMoonshineSetInstanceVisibility(c, h, false); }   // hydra.zig:495-497
void HdMoonshineSetInstanceTransform(HdMoonshine* c, InstanceHandle h, Mat3x4 t) {   // hydra.zig:499-505: the TLAS is updated at the next render
    LOCK(c);
    if (h >= c->instances.size()) return;
    InstanceH& in = c->instances[h];
    memcpy(&in.transform, &t, sizeof(m34));
    // in place when the edit leaves the scene's structure alone: the instance is and stays a TLAS leaf of its own (not part of the merged world BLAS, which holds
    // the identity-transform instances), is visible, and none of its geometry is a sampled light (the alias table weighs world-space areas)
    bool in_place = !c->accel_dirty && !c->root_in_blas && h < c->built_in_world.size() && !c->built_in_world[h] && h < c->item_of_instance.size() && c->item_of_instance[h] != MAX_UINT
                    && in.visible && !is_identity(in.transform) && finite_transform(in.transform);
    for (const GeometryRec& g : in.geos) if (g.sampled) in_place = false;
    if (in_place) c->transform_edits.push_back(h); else c->accel_dirty = true;
    c->clear_all_sensors();
}

// Accel.recordUpdateSingleMaterial (Accel.zig:609-628) as the online editor calls it (online/main.zig:229-233): geometry `geometry_index` of `instance`
// (the reference's flat geometry index = the instance's custom index + geometry_index) gets another material from the next render on.  The sensors are NOT
// cleared: the reference's caller does that (online/main.zig:231).
int MsneSetGeometryMaterial(HdMoonshine* c, InstanceHandle h, uint32_t geometry_index, MaterialHandle m) {
    LOCK(c);
    if (h >= c->instances.size() || geometry_index >= c->instances[h].geos.size()) { c->fail("geometry material: unknown instance or geometry"); return -1; }
    if (m >= c->materials.size()) { c->fail("geometry material: unknown material"); return -2; }   // (online/main.zig:229 checks the same bound)
    c->instances[h].geos[geometry_index].material = m;
    if (!c->accel_dirty) c->geometry_edits.emplace_back((uint32_t)h, geometry_index);
    return 0;
}

int MsneSetPipeline(HdMoonshine* c, const MsnePipelineOpts* o) {
    LOCK(c);
    if (!o || o->samples_per_run == 0) { c->fail("pipeline: samples_per_run must be >= 1"); return -1; }
    if (o->env_samples_per_bounce > 64 || o->mesh_samples_per_bounce > 64) { c->fail("pipeline: at most 64 env and 64 mesh light samples per bounce"); return -2; }
    if (o->max_bounces > 65000) { c->fail("pipeline: max_bounces too large"); return -1; }
    c->opts = PipelineOpts{ o->samples_per_run, o->max_bounces, o->env_samples_per_bounce, o->mesh_samples_per_bounce, o->flip_image, o->indexed_attributes, o->two_component_normal_texture };
    c->clear_all_sensors();                                   // hydra.zig:365-372
    return 0;
}
int MsneGetPipeline(const HdMoonshine* c, MsnePipelineOpts* o) {
    if (!c || !o) return -1;
    *o = MsnePipelineOpts{ c->opts.samples_per_run, c->opts.max_bounces, c->opts.env_samples, c->opts.mesh_samples, c->opts.flip_image, c->opts.indexed_attributes, c->opts.two_component_normal_texture };
    return 0;
}
bool HdMoonshineRebuildPipeline(HdMoonshine* c) { LOCK(c); c->clear_all_sensors(); return true; }   // nothing to recompile: constants are kernel arguments

int MsneSetBackground(HdMoonshine* c, const float* rgba, Extent2D e) { LOCK(c); if (!c->bind()) return -1; return c->set_background(rgba, e) ? 0 : -1; }

SensorHandle HdMoonshineCreateSensor(HdMoonshine* c, Extent2D e) {
    LOCK(c);
    if (!c->bind() || e.width == 0 || e.height == 0) { c->fail("sensor: bad extent"); return 0xFFFFFFFFu; }
    SensorH* s = new SensorH(); s->extent = e;
    ShardView& sh = s->shard;
    sh.width = e.width; sh.height = e.height; sh.tile_size = c->cfg.tile_size;
    sh.tiles_x = (e.width + sh.tile_size - 1) / sh.tile_size; sh.tiles_y = (e.height + sh.tile_size - 1) / sh.tile_size;
    sh.shard_index = c->cfg.shard_index; sh.shard_count = c->cfg.shard_count;
    const uint32_t total = sh.tiles_x * sh.tiles_y;
    sh.local_tiles = total > sh.shard_index ? (total - sh.shard_index + sh.shard_count - 1) / sh.shard_count : 0;
    sh.pixels = sh.local_tiles * sh.tile_size * sh.tile_size;
    sh.valid_pixels = 0;
    for (uint32_t k = 0; k < sh.local_tiles; k++) {
        const uint32_t t = sh.shard_index + k * sh.shard_count, x0 = (t % sh.tiles_x) * sh.tile_size, y0 = (t / sh.tiles_x) * sh.tile_size;
        sh.valid_pixels += std::min(sh.tile_size, e.width - x0) * std::min(sh.tile_size, e.height - y0);
    }
    const size_t npix = (size_t)e.width * e.height;
    // the packed film is padded to the largest shard so that gathers move equal-sized buffers
    const size_t padded = (size_t)((total + sh.shard_count - 1) / sh.shard_count) * sh.tile_size * sh.tile_size;
    bool ok = s->film_packed.alloc(padded) && s->color.alloc(padded) && s->film_full.alloc(npix)
           && hipHostMalloc((void**)&s->host, npix * 16, hipHostMallocDefault) == hipSuccess;
    if (ok) ok = hipMemsetAsync(s->film_packed.p, 0, std::max<size_t>(padded, 1) * 16, c->stream) == hipSuccess && hipMemsetAsync(s->film_full.p, 0, npix * 16, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
    if (!ok) { if (s->host) (void)hipHostFree(s->host); delete s; c->fail("out of memory (sensor)"); return 0xFFFFFFFFu; }
    memset(s->host, 0, npix * 16);
    c->sensors.push_back(s);
    return (SensorHandle)c->sensors.size() - 1;
}
float* HdMoonshineGetSensorData(const HdMoonshine* c, SensorHandle s) { return s < c->sensors.size() ? (float*)c->sensors[s]->host : nullptr; }
LensHandle HdMoonshineCreateLens(HdMoonshine* c, Lens l) { LOCK(c); c->lenses.push_back(l); return (LensHandle)c->lenses.size() - 1; }
void HdMoonshineSetLens(HdMoonshine* c, LensHandle h, Lens l) { LOCK(c); if (h >= c->lenses.size()) return; c->lenses[h] = l; c->clear_all_sensors(); }
void MsneClearSensor(HdMoonshine* c, SensorHandle s) { LOCK(c); if (s < c->sensors.size()) c->sensors[s]->sample_count = 0; }
uint32_t MsneGetSampleCount(const HdMoonshine* c, SensorHandle s) { return s < c->sensors.size() ? c->sensors[s]->sample_count : 0; }

int MsneReserve(HdMoonshine* c, SensorHandle sh, uint32_t launches) {
    LOCK(c);
    if (!c->bind() || sh >= c->sensors.size()) return -1;
    const size_t P = c->sensors[sh]->shard.pixels, per_launch = P * (size_t)c->opts.samples_per_run;
    if (per_launch == 0) return 0;
    const size_t budget = c->inflight_budget();
    const size_t nb = per_launch <= budget ? std::min<size_t>(std::max<size_t>(1, budget / per_launch), std::max<uint32_t>(launches, 1)) : 1;
    const size_t n = per_launch <= budget ? per_launch * nb : P * std::max<size_t>(1, budget / P);
    const uint32_t nbu = (uint32_t)nb;
    const int K = (per_launch > budget || nbu < 2 || per_launch * nbu >= c->single_pipe_paths) ? 1 : (int)std::min<uint32_t>((uint32_t)c->n_pipes, nbu);
    const size_t per_pipe = per_launch <= budget ? per_launch * ((nbu + K - 1) / K) : n;
    return c->ensure_wavefront(per_pipe, n, K) ? 0 : -1;
}
int MsneSetMaxInflight(HdMoonshine* c, uint64_t paths) {
    LOCK(c);
    if (paths == 0) return -1;
    c->max_inflight = (size_t)paths;
    return 0;
}
uint64_t MsneGetMaxInflight(const HdMoonshine* c) { return c ? (uint64_t)c->max_inflight : 0; }
int MsneRender(HdMoonshine* c, SensorHandle s, LensHandle l, uint32_t launches, int readback) {
    LOCK(c);
    if (!c->bind()) return -1;
    return c->render(s, l, launches, readback != 0) ? 0 : -1;
}
bool HdMoonshineRender(HdMoonshine* c, SensorHandle s, LensHandle l) { return MsneRender(c, s, l, 1, 1) == 0; }

uint64_t MsneGetShardTileCount(const HdMoonshine* c, SensorHandle s) { return s < c->sensors.size() ? c->sensors[s]->shard.local_tiles : 0; }
void* MsneGetPackedFilmDevicePtr(const HdMoonshine* c, SensorHandle s) { return s < c->sensors.size() ? (void*)c->sensors[s]->film_packed.p : nullptr; }
uint64_t MsneGetPackedFilmStride(const HdMoonshine* c, SensorHandle sh) {
    if (!c || sh >= c->sensors.size()) return 0;
    const ShardView& v = c->sensors[sh]->shard;
    const uint32_t total = v.tiles_x * v.tiles_y;
    return (uint64_t)((total + v.shard_count - 1) / v.shard_count) * v.tile_size * v.tile_size;
}
int MsneUnpackGatheredFilm(HdMoonshine* c, SensorHandle sh, const void* gathered, uint32_t shard_count) {
    LOCK(c);
    if (!c->bind() || sh >= c->sensors.size() || !gathered) return -1;
    SensorH* s = c->sensors[sh];
    if (shard_count != s->shard.shard_count) { c->fail("unpack: shard_count mismatch"); return -1; }
    const uint32_t total = s->shard.tiles_x * s->shard.tiles_y;
    const size_t stride = (size_t)((total + shard_count - 1) / shard_count) * s->shard.tile_size * s->shard.tile_size;   // max tiles per shard, padded
    launch_unpack_film(c->stream, 1024, s->shard, (const float4*)gathered, shard_count, 0, stride, s->film_full.p);
    if (hipMemcpyAsync(s->host, s->film_full.p, (size_t)s->extent.width * s->extent.height * 16, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { c->fail("unpack failed"); return -1; }
    return 0;
}

int MsneGetStats(const HdMoonshine* cc, MsneStats* out) {
    HdMoonshine* c = const_cast<HdMoonshine*>(cc);
    LOCK(c);
    if (!out || !c->bind()) return -1;
    *out = c->stats;
    out->closest_rays = out->shadow_rays = out->samples = 0;
    for (auto& pp : c->pipes) {
        if (!pp.totals.p) continue;
        Totals h{};
        if (hipMemcpy(&h, pp.totals.p, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        out->closest_rays += h.closest_rays; out->shadow_rays += h.shadow_rays; out->samples += h.samples;
    }
    return 0;
}
void MsneResetStats(HdMoonshine* c) {
    LOCK(c);
    if (!c->bind()) return;
    c->stats = MsneStats{};
    for (auto& pp : c->pipes) if (pp.totals.p) (void)hipMemset(pp.totals.p, 0, sizeof(Totals));
    if (c->d_trace_stats.p) (void)hipMemset(c->d_trace_stats.p, 0, 352);
}

// ---- diagnostics used by the parity tests (no reference equivalent) ----
void MsneSetProfiling(HdMoonshine* c, int kernel_events, int traversal_counters) {
    LOCK(c); c->profile = kernel_events != 0; c->trace_stats = traversal_counters != 0;
    // kernel_events == 2: time the kernels in stream order (k_trace_shadow(b) is not overlapped with k_trace_closest(b+1)), so that the per-kernel
    // durations are exclusive; 0 / 1 restore the context's own choice ($MSNE_SERIAL or the batch-size rule)
    if (kernel_events == 2) { if (c->serial_saved == -2) c->serial_saved = c->serial_mode; c->serial_mode = 1; }
    else if (c->serial_saved != -2) { c->serial_mode = c->serial_saved; c->serial_saved = -2; }
}
void MsneSetBuildQuality(HdMoonshine* c, int prefer_fast_trace) { LOCK(c); c->fast_builds = prefer_fast_trace == 0; }   // takes effect at the next (re)build
// One wave spins for ~0.2 ms and reads both of the chip's counters around the spin: s_memtime counts shader-engine clocks, s_memrealtime the constant 100 MHz
// reference.  Their ratio is the shader clock WHILE whatever else runs on the GPU runs (the probe has a stream of its own and takes one wave slot).
__global__ void k_clock_probe(unsigned long long* out, uint32_t spins) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = (float)threadIdx.x;
    for (uint32_t i = 0; i < spins; i++) { x = __builtin_fmaf(x, 1.0000001f, 1e-9f); asm volatile("" : "+v"(x)); }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
double MsneProbeClockGhz(int device) {
    struct Probe { hipStream_t s = nullptr; unsigned long long* d = nullptr; };
    // one stream and 32 bytes per device that was ever probed, kept for the life of the process ON PURPOSE: a static destructor would call into HIP after the runtime
    // has begun to shut down (the order of the two at exit is not ours), which is worse than 32 bytes
    struct Probes { std::mutex mu; std::map<int, Probe> of; };
    static Probes probes;
    std::lock_guard<std::mutex> g(probes.mu);
    if (hipSetDevice(device) != hipSuccess) return -1.0;
    Probe& p = probes.of[device];
    if (!p.s) {
        if (hipStreamCreateWithFlags(&p.s, hipStreamNonBlocking) != hipSuccess) { p.s = nullptr; return -1.0; }
        if (hipMalloc(&p.d, 32) != hipSuccess) { (void)hipStreamDestroy(p.s); p.s = nullptr; p.d = nullptr; return -1.0; }   // (nothing is kept of a probe that is half made)
    }
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, p.s, p.d, 60000u);
    if (hipGetLastError() != hipSuccess) return -1.0;
    unsigned long long h[3] = { 0, 0, 0 };
    if (hipMemcpyAsync(h, p.d, 24, hipMemcpyDeviceToHost, p.s) != hipSuccess || hipStreamSynchronize(p.s) != hipSuccess || h[1] == 0) return -1.0;
    return (double)h[0] / (double)h[1] * 0.1;
}
void MsneGetAccelStats(HdMoonshine* c, uint64_t out[2]) { LOCK(c); out[0] = c->n_rebuilds; out[1] = c->n_tlas_updates; }   // acceleration-structure rebuilds, in-place TLAS updates
uint64_t MsneGetTexelPoolBytes(HdMoonshine* c) {   // bytes of texels resident in HBM (after the next upload: what has been created so far)
    LOCK(c);
    uint64_t n = (uint64_t)c->texels_end * 16u;
    for (const auto& t : c->textures) if (!t.on_device) n += t.raw.size();
    return n;
}
int MsneGetTraversalCounters(HdMoonshine* c, uint64_t out[20]) {   // [0..3] closest {node visits, tri tests}, shadow {..}; [4..11] closest wave-cycle profile, [12..19] shadow
    LOCK(c);
    if (!c->bind() || !c->d_trace_stats.p) { for (int i = 0; i < 20; i++) out[i] = 0; return 0; }
    unsigned long long h[20];
    if (hipMemcpy(h, c->d_trace_stats.p, 160, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (int i = 0; i < 20; i++) out[i] = h[i];
    return 0;
}
int MsneGetTraversalLaneUse(HdMoonshine* c, uint64_t out[24]) {   // [0..11] closest, [12..23] shadow: the lane-use counters of trace_wave_loop (STATS instantiation)
    LOCK(c);
    for (int i = 0; i < 24; i++) out[i] = 0;
    if (!c->bind() || !c->d_trace_stats.p) return 0;
    unsigned long long h[24];
    if (hipMemcpy(h, c->d_trace_stats.p + 20, 192, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (int i = 0; i < 24; i++) out[i] = h[i];
    return 0;
}
// rays: 7 floats each (origin, direction, tmax); out_ids 4 per ray {hit, instance, geometry, primitive}; out_tuv 3 per ray
int MsneTraceRays(HdMoonshine* c, const float* rays, uint32_t n, int any_hit, uint32_t* out_ids, float* out_tuv) {
    LOCK(c);
    c->origin_needed = 0.0f;
    for (uint32_t i = 0; rays && i < n; i++) c->need_origin(rays[7 * (size_t)i], rays[7 * (size_t)i + 1], rays[7 * (size_t)i + 2]);
    if (!c->bind() || !c->ensure_scene() || !c->ensure_wavefront(1, 1, 1)) return -1;
    if (n == 0) return 0;
    DevBuf<float> dr; DevBuf<uint32_t> di; DevBuf<float> dt;
    if (!dr.alloc(7 * (size_t)n) || !di.alloc(4 * (size_t)n) || !dt.alloc(3 * (size_t)n)) { c->fail("out of device memory (probe)"); return -1; }
    if (hipMemcpyAsync(dr.p, rays, 28 * (size_t)n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return -1;
    if (hipMemsetAsync(di.p, 0, 16 * (size_t)n, c->stream) != hipSuccess || hipMemsetAsync(dt.p, 0, 12 * (size_t)n, c->stream) != hipSuccess) return -1;
    if (hipMemsetAsync(&c->pipes[0].counters.p->head_closest, 0, 4, c->stream) != hipSuccess) return -1;
    launch_trace_probe(c->stream, c->trace_grid, c->scene_view(), dr.p, n, any_hit, &c->pipes[0].counters.p->head_closest, di.p, dt.p, c->pipes[0].spill.p, c->d_overflow.p, c->refill);
    if (hipMemcpyAsync(out_ids, di.p, 16 * (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipMemcpyAsync(out_tuv, dt.p, 12 * (size_t)n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(c->stream) != hipSuccess) { c->fail("probe failed"); return -1; }
    return c->check_overflow() ? 0 : -1;
}
// the last render made with kernel events on (MsneSetProfiling): every k_trace_closest (kind 0) / k_trace_shadow (1) / k_shade (2) launch in issue order — closest(b),
// shade(b), shadow(b) for b = 0, 1, ... per batch and pipe — with its HIP-event duration: what a bounce of a shard costs, per rank (bench.py per_rank)
int MsneGetLaunchTimes(HdMoonshine* c, int32_t* kinds, float* ms, uint32_t max_launches) {
    LOCK(c);
    const uint32_t n = (uint32_t)std::min<size_t>(max_launches, c->launch_times.size());
    for (uint32_t i = 0; i < n; i++) { if (kinds) kinds[i] = c->launch_times[i].first; if (ms) ms[i] = c->launch_times[i].second; }
    return (int)n;
}
// queue lengths of the last batch traced on pipe 0, per bounce: {paths, of which zombies, shadow entries, shadow rays traced}
int MsneGetBounceCounters(HdMoonshine* c, uint32_t* out, uint32_t max_bounces) {
    LOCK(c);
    if (!c->bind() || !out || !c->pipes[0].counters.p) return -1;
    const uint32_t n = (uint32_t)std::min<size_t>(max_bounces, c->pipes[0].counters.n);
    std::vector<BounceCounters> h(n);
    if (hipMemcpy(h.data(), c->pipes[0].counters.p, n * sizeof(BounceCounters), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (uint32_t b = 0; b < n; b++) {   // a queue's entries = the sum over its sub-queues (the holes between them are nobody's entries)
        uint32_t np = 0, nsh = 0;
        for (uint32_t k = 0; k < QUEUE_SUBS; k++) { np += h[b].sub[k].n_paths; nsh += h[b].sub[k].n_shadow; }
        out[4 * b] = np; out[4 * b + 1] = h[b].zombies; out[4 * b + 2] = nsh; out[4 * b + 3] = h[b].n_shadow_traced;
    }
    return (int)n;
}
// batch probe of the device shading functions (material / light / mapping code of k_shade), see k_shade_probe
int MsneShadeProbe(HdMoonshine* c, int fn, const float* in, uint32_t n, float* out) {
    LOCK(c);
    uint32_t win = 0, wout = 0;
    if (!c->bind() || !in || !out || !shade_probe_widths(fn, win, wout)) { c->fail("shade probe: bad arguments"); return -1; }
    if (n == 0) return 0;
    if (fn == 17) {   // texture probe: the scene's textures must be on the device, and every index must name one
        if (c->textures_dirty && !c->upload_textures()) return -1;
        for (uint32_t i = 0; i < n; i++) if (!(in[3 * (size_t)i] >= 0.0f && in[3 * (size_t)i] < (float)c->textures.size())) { c->fail("shade probe: unknown texture"); return -1; }
    }
    std::vector<float> cams;
    if (fn == 20) {   // camera probe: the caller hands over lenses; the constants MsneRender would derive from them (make_camera) go to the device
        cams.resize((size_t)n * win);
        for (uint32_t i = 0; i < n; i++) {
            const float* a = in + 18 * (size_t)i; float* o = &cams[(size_t)i * win];
            if (!(a[12] >= 1.0f && a[13] >= 1.0f)) { c->fail("shade probe: camera record without an extent"); return -1; }
            Lens lens{}; lens.origin = F32x3{ a[0], a[1], a[2] }; lens.forward = F32x3{ a[3], a[4], a[5] }; lens.up = F32x3{ a[6], a[7], a[8] };
            lens.vfov = a[9]; lens.aperture = a[10]; lens.focus_distance = a[11];
            const CameraConsts k = make_camera(lens, (uint32_t)a[12], (uint32_t)a[13]);
            const f3 v[6] = { k.origin, k.u, k.v, k.horizontal, k.vertical, k.llc };
            for (int j = 0; j < 6; j++) { o[3 * j] = v[j].x; o[3 * j + 1] = v[j].y; o[3 * j + 2] = v[j].z; }
            o[18] = k.aperture; o[19] = a[14]; o[20] = a[15]; o[21] = a[16]; o[22] = a[17];
        }
        in = cams.data();
    }
    DevBuf<float> di, dout;
    if (!di.alloc((size_t)n * win) || !dout.alloc((size_t)n * wout)) { c->fail("out of device memory (probe)"); return -1; }
    if (hipMemcpyAsync(di.p, in, (size_t)n * win * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) return -1;
    launch_shade_probe(c->stream, c->scene_view(), fn, di.p, n, dout.p);
    if (hipMemcpyAsync(out, dout.p, (size_t)n * wout * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { c->fail("shade probe failed"); return -1; }
    return 0;
}
int MsnePick(HdMoonshine* c, SensorHandle sensor, LensHandle lens, F32x2 nc, MsneClickData* out) {
    if (!c || !out) return -1;
    f3 O, D;
    {
        LOCK(c);
        if (sensor >= c->sensors.size() || lens >= c->lenses.size()) { c->fail("pick: bad sensor or lens handle"); return -1; }
        const SensorH* s = c->sensors[sensor];
        const CameraConsts cam = make_camera(c->lenses[lens], s->extent.width, s->extent.height);
        float v = nc.y; v -= 1.0f; v *= -1.0f;                       // input.hlsl:46-48
        camera_generate_ray(cam, F2(nc.x, v), F2(0.0f, 0.0f), O, D);
    }
    const float ray[7] = { O.x, O.y, O.z, D.x, D.y, D.z, INFINITY_F };
    uint32_t ids[4] = { 0, 0, 0, 0 }; float tuv[3] = { 0, 0, 0 };
    if (MsneTraceRays(c, ray, 1, 0, ids, tuv) != 0) return -1;
    out->instance_index = ids[0] ? (int32_t)ids[1] : -1; out->geometry_index = ids[0] ? ids[2] : 0u; out->primitive_index = ids[0] ? ids[3] : 0u;
    out->barycentrics = F32x2{ ids[0] ? tuv[1] : 0.0f, ids[0] ? tuv[2] : 0.0f };
    return 0;
}
uint32_t MsneGetEnvSize(const HdMoonshine* c) { return c->env.size; }
int MsneReadEnv(HdMoonshine* c, float* rgb_out /*S*S*4*/, float* lum_out /*whole pyramid*/) {
    LOCK(c);
    if (!c->bind()) return -1;
    const uint32_t S = c->env.size; size_t total = 0;
    for (uint32_t l = 0; l < c->env.mip_count; l++) total += (size_t)(S >> l) * (S >> l);
    if (rgb_out && hipMemcpy(rgb_out, c->d_env_rgb.p, (size_t)S * S * 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (lum_out && hipMemcpy(lum_out, c->d_env_lum.p, total * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return 0;
}
uint32_t MsneGetAliasTable(HdMoonshine* c, void* out, uint32_t max_entries) {   // entries of 20 B, entry 0 = header
    LOCK(c);
    if (!c->bind() || !c->ensure_scene()) return 0;
    const uint32_t n = (uint32_t)c->h_alias.size();
    if (out) memcpy(out, c->h_alias.data(), (size_t)std::min(n, max_entries) * sizeof(AliasEntry));
    return n;
}
// BVH dump for structural validation: copies nodes / triangles / tlas items to the host
int MsneReadBvh(HdMoonshine* c, void* nodes_out, uint32_t* node_count, void* tris_out, uint32_t* tri_count, uint32_t* tlas_root, uint32_t* tlas_items_out, uint32_t* tlas_item_count) {
    LOCK(c);
    if (!c->bind() || !c->ensure_scene()) return -1;
    uint32_t cnt[3];
    if (hipMemcpy(cnt, c->d_build_counters.p, 12, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (node_count) { if (nodes_out && *node_count >= cnt[0] && hipMemcpy(nodes_out, c->d_nodes.p, (size_t)cnt[0] * sizeof(Node8), hipMemcpyDeviceToHost) != hipSuccess) return -1; *node_count = cnt[0]; }
    if (tri_count) { if (tris_out && *tri_count >= cnt[1] && hipMemcpy(tris_out, c->d_tris.p, (size_t)cnt[1] * sizeof(TriRec), hipMemcpyDeviceToHost) != hipSuccess) return -1; *tri_count = cnt[1]; }
    if (tlas_item_count) { if (tlas_items_out && *tlas_item_count >= cnt[2] && cnt[2] && hipMemcpy(tlas_items_out, c->d_tlas_items.p, (size_t)cnt[2] * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1; *tlas_item_count = cnt[2]; }
    if (tlas_root) *tlas_root = c->tlas_root;
    return 0;
}

}  // extern "C"
