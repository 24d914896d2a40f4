// group.hip — native multi-GPU rendering behind the C ABI (include/moonshine_amd.h, "MsneGroup*").
// The reference renders on ONE device (engine/core/VulkanContext.zig:313-326); north_star shards the image's tiles over the
// GPUs of a node.  A group is one HdMoonshine context per GPU (scene replicated, tile t -> member t mod n), one host thread per
// member for the calls that take time, and ONE collective in the data path: ncclGather (RCCL over xGMI) of the members' packed
// films to member 0, where k_unpack_film scatters them into the row-major image (offline/main.zig:80-203 is the caller's shape:
// load, create pipeline, trace spp launches, copy the image to the host, write the EXR).
// RCCL is loaded with dlopen at the first gather that needs it: the library itself carries no link dependency on it, and a
// process that already holds another copy (PyTorch bundles one) keeps the two apart.  Several members on the SAME GPU (tests,
// one-GPU boxes) cannot form an RCCL communicator; their gather is a device-to-device copy.
#include "../../include/moonshine_amd.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {

// the five RCCL entry points used, with rccl.h's signatures (ncclFloat = 7, ncclSuccess = 0)
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Gather)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        const char* env = getenv("MSNE_RCCL_PATH");
        const char* names[] = { env, "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so" };
        for (const char* n : names) { if (!n) continue; lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Gather = (decltype(Gather))dlsym(lib, "ncclGather"); GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Gather) { err = "librccl.so lacks ncclCommInitAll / ncclGather"; dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};

}  // namespace

struct MsneGroup {
    std::vector<HdMoonshine*> ctx;
    std::vector<int> dev;
    bool distinct = true;                    // every member on its own GPU: the gather can be an RCCL collective
    Rccl rccl; std::vector<void*> comms;     // communicators, created at the first gather
    std::vector<hipStream_t> streams;        // one gather stream per member
    void* gathered = nullptr; size_t gathered_bytes = 0;   // n padded packed films, on member 0's GPU
    std::string last_error; std::string transport = "none";
    std::mutex mutex;
    uint64_t gathers = 0; double gather_ms = 0.0;
    void fail(const std::string& m) { last_error = m; if (getenv("MSNE_VERBOSE")) fprintf(stderr, "moonshine_amd group: %s\n", m.c_str()); }
};

static thread_local std::string g_group_create_error;

// run fn(i) for every member on its own host thread (the calls below block: BVH builds, renders, PNG decoding)
template <class F> static bool for_each_member(MsneGroup* g, F fn) {
    const size_t n = g->ctx.size();
    std::vector<int> ok(n, 0);
    if (n == 1) { ok[0] = fn(0) ? 1 : 0; }
    else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < n; i++) th.emplace_back([&, i] { ok[i] = fn(i) ? 1 : 0; });
        for (auto& t : th) t.join();
    }
    for (size_t i = 0; i < n; i++) if (!ok[i]) { const char* e = MsneGetLastError(g->ctx[i]); g->fail("member " + std::to_string(i) + ": " + (e && *e ? e : MsneGetIoError())); return false; }
    return true;
}

// $MSNE_GROUP_FORCE_RCCL=1: a one-member group also goes through the gather (ncclCommInitAll with one device) — the only way to run the
// RCCL leg on a one-GPU machine; $MSNE_GROUP_NO_RCCL=1: always copy
static bool env_flag(const char* name) { const char* e = getenv(name); return e && atoi(e) != 0; }
static bool gathers(const MsneGroup* g) { return g->ctx.size() > 1 || env_flag("MSNE_GROUP_FORCE_RCCL"); }

static bool gather_and_unpack(MsneGroup* g, SensorHandle sensor) {
    const size_t n = g->ctx.size();
    const uint64_t stride = MsneGetPackedFilmStride(g->ctx[0], sensor);     // float4 per member, padded to the largest shard
    if (stride == 0) { g->fail("gather: bad sensor"); return false; }
    const size_t bytes = (size_t)stride * 16;
    if (hipSetDevice(g->dev[0]) != hipSuccess) { g->fail("hipSetDevice failed"); return false; }
    if (g->gathered_bytes < bytes * n) {
        if (g->gathered) (void)hipFree(g->gathered);
        g->gathered = nullptr; g->gathered_bytes = 0;
        if (hipMalloc(&g->gathered, bytes * n) != hipSuccess) { g->fail("out of device memory (gathered films)"); return false; }
        g->gathered_bytes = bytes * n;
    }
    if (g->streams.empty()) {   // all or nothing: a half-built set would make the next call run on null streams of the wrong device
        std::vector<hipStream_t> st(n, nullptr);
        bool ok = true;
        for (size_t i = 0; i < n && ok; i++) ok = hipSetDevice(g->dev[i]) == hipSuccess && hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) == hipSuccess;
        if (!ok) {
            for (size_t i = 0; i < n; i++) if (st[i]) { (void)hipSetDevice(g->dev[i]); (void)hipStreamDestroy(st[i]); }
            g->fail("cannot create gather streams"); return false;
        }
        g->streams.swap(st);
    }
    struct EventPair {   // destroyed on every way out
        hipEvent_t e0 = nullptr, e1 = nullptr; int dev;
        explicit EventPair(int d) : dev(d) {}
        ~EventPair() { (void)hipSetDevice(dev); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev(g->dev[0]);
    hipEvent_t& e0 = ev.e0; hipEvent_t& e1 = ev.e1;
    (void)hipSetDevice(g->dev[0]);
    if (hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) (void)hipEventRecord(e0, g->streams[0]);
    // communicators that saw a failed collective may be unusable: drop them, the next gather re-initialises
    auto drop_comms = [&] { for (size_t i = 0; i < g->comms.size(); i++) if (g->comms[i] && g->rccl.CommDestroy) { (void)hipSetDevice(g->dev[i]); (void)g->rccl.CommDestroy(g->comms[i]); } g->comms.clear(); };
    const bool use_rccl = g->distinct && !env_flag("MSNE_GROUP_NO_RCCL");
    if (use_rccl) {
        if (g->comms.empty()) {
            if (!g->rccl.load(g->last_error)) return false;
            g->comms.resize(n, nullptr);
            const int r = g->rccl.CommInitAll(g->comms.data(), (int)n, g->dev.data());
            if (r != 0) { g->comms.clear(); g->fail(std::string("ncclCommInitAll: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(r) : "error")); return false; }
        }
        // ONE collective: every member sends its packed film, member 0 receives them back to back (7 peers over 7 xGMI links in parallel)
        int r = g->rccl.GroupStart();
        for (size_t i = 0; i < n && r == 0; i++) {
            (void)hipSetDevice(g->dev[i]);
            r = g->rccl.Gather(MsneGetPackedFilmDevicePtr(g->ctx[i], sensor), g->gathered, (size_t)stride * 4, 7 /* ncclFloat */, 0, g->comms[i], g->streams[i]);
        }
        const int r2 = g->rccl.GroupEnd();
        if (r != 0 || r2 != 0) { g->fail(std::string("ncclGather: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(r ? r : r2) : "error")); drop_comms(); return false; }
        for (size_t i = 0; i < n; i++) { (void)hipSetDevice(g->dev[i]); if (hipStreamSynchronize(g->streams[i]) != hipSuccess) { g->fail("gather failed"); drop_comms(); return false; } }
        g->transport = "rccl";
    } else {
        (void)hipSetDevice(g->dev[0]);
        for (size_t i = 0; i < n; i++) {
            const void* src = MsneGetPackedFilmDevicePtr(g->ctx[i], sensor);
            const hipError_t e = g->dev[i] == g->dev[0] ? hipMemcpyAsync((char*)g->gathered + i * bytes, src, bytes, hipMemcpyDeviceToDevice, g->streams[0])
                                                        : hipMemcpyPeerAsync((char*)g->gathered + i * bytes, g->dev[0], src, g->dev[i], bytes, g->streams[0]);
            if (e != hipSuccess) { g->fail(std::string("film copy failed: ") + hipGetErrorString(e)); return false; }
        }
        if (hipStreamSynchronize(g->streams[0]) != hipSuccess) { g->fail("film copy failed"); return false; }
        g->transport = "copy";
    }
    (void)hipSetDevice(g->dev[0]);
    if (e0 && e1) { float ms = 0.0f; (void)hipEventRecord(e1, g->streams[0]); if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) g->gather_ms += ms; }
    g->gathers++;
    if (MsneUnpackGatheredFilm(g->ctx[0], sensor, g->gathered, (uint32_t)n) != 0) { g->fail(std::string("unpack: ") + MsneGetLastError(g->ctx[0])); return false; }
    return true;
}

extern "C" {

MsneGroup* MsneGroupCreate(const int32_t* devices, uint32_t n, uint32_t tile_size) {
    if (n == 0 || n > 64) { g_group_create_error = "group: 1..64 members"; return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_group_create_error = "no HIP device available: libmoonshine_amd requires an MI355X (gfx950) GPU"; return nullptr; }
    MsneGroup* g = new (std::nothrow) MsneGroup();
    if (!g) return nullptr;
    std::set<int> seen;
    for (uint32_t i = 0; i < n; i++) {
        const int d = devices ? devices[i] : (int)(i % (uint32_t)ndev);
        if (d < 0 || d >= ndev) { g_group_create_error = "group: HIP device " + std::to_string(d) + " does not exist"; MsneGroupDestroy(g); return nullptr; }
        if (!seen.insert(d).second) g->distinct = false;
        MsneConfig cfg{ d, tile_size, i, n };
        HdMoonshine* c = MsneCreate(&cfg);
        if (!c) { g_group_create_error = std::string("group member: ") + MsneGetLastError(nullptr); MsneGroupDestroy(g); return nullptr; }
        g->ctx.push_back(c); g->dev.push_back(d);
    }
    for (uint32_t i = 0; i < n; i++) {   // members that share a GPU share its memory: split the wavefront-state budget between them
        uint32_t same = 0;
        for (uint32_t k = 0; k < n; k++) same += g->dev[k] == g->dev[i] ? 1u : 0u;
        if (same > 1) (void)MsneSetMaxInflight(g->ctx[i], std::max<uint64_t>(1u << 20, MsneGetMaxInflight(g->ctx[i]) / same));
    }
    return g;
}

void MsneGroupDestroy(MsneGroup* g) {
    if (!g) return;
    for (size_t i = 0; i < g->comms.size(); i++) if (g->comms[i]) { (void)hipSetDevice(g->dev[i]); (void)g->rccl.CommDestroy(g->comms[i]); }
    for (size_t i = 0; i < g->streams.size(); i++) if (g->streams[i]) { (void)hipSetDevice(g->dev[i]); (void)hipStreamDestroy(g->streams[i]); }
    if (g->gathered) { (void)hipSetDevice(g->dev[0]); (void)hipFree(g->gathered); }
    for (HdMoonshine* c : g->ctx) HdMoonshineDestroy(c);
    delete g;
}

uint32_t MsneGroupSize(const MsneGroup* g) { return g ? (uint32_t)g->ctx.size() : 0; }
HdMoonshine* MsneGroupContext(MsneGroup* g, uint32_t i) { return g && i < g->ctx.size() ? g->ctx[i] : nullptr; }
const char* MsneGroupGetLastError(const MsneGroup* g) { return g ? g->last_error.c_str() : g_group_create_error.c_str(); }
const char* MsneGroupTransport(const MsneGroup* g) { return g ? g->transport.c_str() : "none"; }

// scene replication: the same call on every member, one host thread each (handles come out equal: members start identical)
int MsneGroupLoadGlb(MsneGroup* g, const char* path, MsneGlbInfo* info) {
    if (!g || !path) return -1;
    std::lock_guard<std::mutex> lock(g->mutex);
    std::vector<MsneGlbInfo> infos(g->ctx.size());
    if (!for_each_member(g, [&](size_t i) { return MsneLoadGlb(g->ctx[i], path, &infos[i]) == 0; })) return -1;
    for (size_t i = 1; i < infos.size(); i++) if (memcmp(&infos[i], &infos[0], sizeof(MsneGlbInfo)) != 0) { g->fail("group members diverged while loading the scene"); return -1; }
    if (info) *info = infos[0];
    return 0;
}
int MsneGroupSetBackgroundExr(MsneGroup* g, const char* path) {
    if (!g || !path) return -1;
    std::lock_guard<std::mutex> lock(g->mutex);
    return for_each_member(g, [&](size_t i) { return MsneSetBackgroundExr(g->ctx[i], path) == 0; }) ? 0 : -1;
}
int MsneGroupSetPipeline(MsneGroup* g, const MsnePipelineOpts* o) {
    if (!g) return -1;
    std::lock_guard<std::mutex> lock(g->mutex);
    for (size_t i = 0; i < g->ctx.size(); i++) if (MsneSetPipeline(g->ctx[i], o) != 0) { g->fail(MsneGetLastError(g->ctx[i])); return -1; }
    return 0;
}
int64_t MsneGroupCreateSensor(MsneGroup* g, Extent2D e) {
    if (!g) return -1;
    std::lock_guard<std::mutex> lock(g->mutex);
    int64_t h = -1;
    for (size_t i = 0; i < g->ctx.size(); i++) {
        const SensorHandle s = HdMoonshineCreateSensor(g->ctx[i], e);
        if (s == 0xFFFFFFFFu || (i && (int64_t)s != h)) { g->fail("group: sensor creation failed or members diverged"); return -1; }
        h = s;
    }
    return h;
}

// `launches` back-to-back launches on every member's tiles (concurrently, one host thread per GPU), then the ONE gather and the
// unpack on member 0: HdMoonshineGetSensorData(MsneGroupContext(g, 0), sensor) holds the full image afterwards.
int MsneGroupRender(MsneGroup* g, SensorHandle sensor, LensHandle lens, uint32_t launches) {
    if (!g) return -1;
    std::lock_guard<std::mutex> lock(g->mutex);
    if (!gathers(g)) { if (MsneRender(g->ctx[0], sensor, lens, launches, 1) != 0) { g->fail(MsneGetLastError(g->ctx[0])); return -1; } return 0; }
    if (!for_each_member(g, [&](size_t i) { return MsneRender(g->ctx[i], sensor, lens, launches, 0) == 0; })) return -1;
    return gather_and_unpack(g, sensor) ? 0 : -1;
}

// The `online` front end's frame loop without a window (online/main.zig:287-305,415-416): per frame — clear the sensor once it
// has passed max_sample_count; one launch of samples_per_run while it is below (or always, when max_sample_count == 0); "present";
// then count the frame's samples, clamped to max_sample_count.  Presenting = gathering the members' films and handing the image
// to `present` (the reference blits the sensor to the swapchain every frame); gather_every > 1 presents only every so many frames
// (sharded films are only assembled when somebody looks).  `present` may be NULL.  Returns 0, or -1 / the callback's nonzero value.
int MsneGroupRenderProgressive(MsneGroup* g, SensorHandle sensor, LensHandle lens, uint32_t frames, uint32_t max_sample_count, uint32_t gather_every,
                               MsnePresentFn present, void* user) {
    if (!g) return -1;
    std::lock_guard<std::mutex> lock(g->mutex);
    if (gather_every == 0) gather_every = 1;
    MsnePipelineOpts o;
    if (MsneGetPipeline(g->ctx[0], &o) != 0) return -1;
    const bool single = !gathers(g);
    // the reference counts a frame's samples after it was presented and clamps the count (online/main.zig:415-416); the library
    // counts at launch time, so the clamp and the clear-when-over rule are applied to its counter here
    uint32_t count = MsneGetSampleCount(g->ctx[0], sensor);
    for (uint32_t f = 0; f < frames; f++) {
        if (max_sample_count != 0 && count > max_sample_count) { for (HdMoonshine* c : g->ctx) MsneClearSensor(c, sensor); count = 0; }
        const bool launch = max_sample_count == 0 || count < max_sample_count;
        if (launch && !for_each_member(g, [&](size_t i) { return MsneRender(g->ctx[i], sensor, lens, 1, single && (f + 1) % gather_every == 0) == 0; })) return -1;
        if ((f + 1) % gather_every == 0 || f + 1 == frames) {
            if (!single) { if (!gather_and_unpack(g, sensor)) return -1; }
            else if (!launch || (f + 1) % gather_every != 0) { if (MsneRender(g->ctx[0], sensor, lens, 0, 1) != 0) return -1; }   // readback only
            if (present) { const int r = present(user, f, HdMoonshineGetSensorData(g->ctx[0], sensor), MsneGetSampleCount(g->ctx[0], sensor)); if (r != 0) return r; }
        }
        if (launch) { count += o.samples_per_run; if (max_sample_count != 0 && count > max_sample_count) count = max_sample_count; }
    }
    return 0;
}

int MsneGroupGetStats(MsneGroup* g, MsneStats* out, double* gather_ms, uint64_t* gathers) {
    if (!g || !out) return -1;
    std::lock_guard<std::mutex> lock(g->mutex);
    MsneStats sum{};
    for (HdMoonshine* c : g->ctx) {
        MsneStats s;
        if (MsneGetStats(c, &s) != 0) return -1;
        sum.closest_rays += s.closest_rays; sum.shadow_rays += s.shadow_rays; sum.samples += s.samples; sum.launches = s.launches;
        sum.trace_closest_ms += s.trace_closest_ms; sum.trace_shadow_ms += s.trace_shadow_ms; sum.shade_ms += s.shade_ms;
        sum.render_ms = s.render_ms > sum.render_ms ? s.render_ms : sum.render_ms;     // the job runs at the pace of its slowest member
        sum.trace_closest_launches += s.trace_closest_launches; sum.trace_shadow_launches += s.trace_shadow_launches; sum.shade_launches += s.shade_launches;
    }
    *out = sum;
    if (gather_ms) *gather_ms = g->gather_ms;
    if (gathers) *gathers = g->gathers;
    return 0;
}

}  // extern "C"
