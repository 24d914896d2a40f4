// offline.cpp — the headless renderer of the reference (offline/main.zig:27-203) on top of the C ABI only:
//     offline <in.glb> <skybox.exr> <out.exr> [spp=16] [--width W --height H --max-bounces N --env-samples N --mesh-samples N]
//             [--gpus N | --devices a,b,..]   tiles sharded over N GPUs, one RCCL gather of the films (BASELINE configs[3])
//             [--progressive FRAMES --max-sample-count N --present-every K]   the `online` frame loop without a window
// Same positional arguments, extension checks, defaults (1280x720, 16 spp, max_bounces 1024, one env + one mesh light
// sample per bounce, samples_per_run 1 — offline/main.zig:41-50,106-111) and the same timing lines (:59-76).
#include "../../include/moonshine_amd.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" const char* MsneGetIoError(void);

static bool has_ext(const std::string& p, const char* e) { const size_t n = strlen(e); return p.size() >= n && p.compare(p.size() - n, n, e) == 0; }

struct IntervalLogger {
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void log(const char* what) {
        const auto now = std::chrono::steady_clock::now();
        printf("%.3f seconds to %s\n", std::chrono::duration<double>(now - last).count(), what);
        last = now;
    }
};

struct ProgressiveLog { IntervalLogger* logger; };
static int present_frame(void* user, uint32_t frame, const float* rgba, uint32_t sample_count) {
    // the headless stand-in for the reference's blit-to-swapchain: report what a viewer would see
    double lum = 0.0; const Extent2D* e = (const Extent2D*)user;
    const size_t n = (size_t)e->width * e->height;
    for (size_t i = 0; i < n; i++) lum += 0.2126 * rgba[4 * i] + 0.7152 * rgba[4 * i + 1] + 0.0722 * rgba[4 * i + 2];
    printf("frame %u: %u samples per pixel, mean luminance %.6f\n", frame, sample_count, lum / (double)n);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: offline <in.glb> <skybox.exr> <out.exr> [spp] [--width W] [--height H] [--max-bounces N] [--env-samples N] [--mesh-samples N]\n"
                                    "               [--gpus N] [--devices a,b,...] [--progressive FRAMES] [--max-sample-count N] [--present-every N]\n"); return 2; }
    const std::string in = argv[1], sky = argv[2], out = argv[3];
    if (!has_ext(in, ".glb")) { fprintf(stderr, "error: OnlySupportsGlbInput\n"); return 2; }
    if (!has_ext(sky, ".exr")) { fprintf(stderr, "error: OnlySupportsExrSkybox\n"); return 2; }
    if (!has_ext(out, ".exr")) { fprintf(stderr, "error: OnlySupportsExrOutput\n"); return 2; }
    uint32_t spp = 16; Extent2D extent{ 1280, 720 };
    MsnePipelineOpts opts{ 1, 1024, 1, 1, 1, 1, 1 };
    uint32_t gpus = 1, progressive = 0, max_sample_count = 0, present_every = 1;
    std::vector<int32_t> devices;
    int a = 4;
    if (a < argc && argv[a][0] != '-') spp = (uint32_t)atoi(argv[a++]);
    for (; a + 1 < argc; a += 2) {
        const std::string k = argv[a]; const uint32_t v = (uint32_t)atoi(argv[a + 1]);
        if (k == "--width") extent.width = v; else if (k == "--height") extent.height = v; else if (k == "--max-bounces") opts.max_bounces = v;
        else if (k == "--env-samples") opts.env_samples_per_bounce = v; else if (k == "--mesh-samples") opts.mesh_samples_per_bounce = v;
        else if (k == "--gpus") gpus = v; else if (k == "--progressive") progressive = v; else if (k == "--max-sample-count") max_sample_count = v;
        else if (k == "--present-every") present_every = v;
        else if (k == "--devices") {   // a comma-separated list of device ordinals
            for (const char* p = argv[a + 1]; *p;) {
                char* end = nullptr;
                const long d = strtol(p, &end, 10);
                if (end == p || (*end != ',' && *end != 0) || d < 0) { fprintf(stderr, "bad --devices list '%s' (expected e.g. 0,1,2)\n", argv[a + 1]); return 2; }
                devices.push_back((int32_t)d);
                p = *end == ',' ? end + 1 : end;
                if (*end == ',' && !*p) { fprintf(stderr, "bad --devices list '%s' (trailing comma)\n", argv[a + 1]); return 2; }
            }
        }
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (!devices.empty()) gpus = (uint32_t)devices.size();
    if (gpus == 0) { fprintf(stderr, "error: --gpus must be at least 1\n"); return 2; }
    IntervalLogger logger;
    // one context per GPU: tiles shard over them, one RCCL gather of the films at the end (the reference is single-device; N = 1 is its shape)
    MsneGroup* group = MsneGroupCreate(devices.empty() ? nullptr : devices.data(), gpus, 0);
    if (!group) { fprintf(stderr, "error: %s\n", MsneGroupGetLastError(nullptr)); return 1; }
    logger.log("set up initial state");
    MsneGlbInfo info;
    if (MsneGroupLoadGlb(group, in.c_str(), &info) != 0) { fprintf(stderr, "error loading %s: %s\n", in.c_str(), MsneGroupGetLastError(group)); return 1; }
    if (MsneGroupSetBackgroundExr(group, sky.c_str()) != 0) { fprintf(stderr, "error loading %s: %s\n", sky.c_str(), MsneGroupGetLastError(group)); return 1; }
    const int64_t sh = MsneGroupCreateSensor(group, extent);
    if (sh < 0) { fprintf(stderr, "error: %s\n", MsneGroupGetLastError(group)); return 1; }
    const SensorHandle sensor = (SensorHandle)sh;
    logger.log("load world");
    if (MsneGroupSetPipeline(group, &opts) != 0) { fprintf(stderr, "error: %s\n", MsneGroupGetLastError(group)); return 1; }
    logger.log("create pipeline");
    if (progressive) {   // the `online` frame loop, headless: `progressive` frames of samples_per_run samples, presented every `present_every` frames
        if (MsneGroupRenderProgressive(group, sensor, info.lens, progressive, max_sample_count, present_every, present_frame, &extent) != 0) { fprintf(stderr, "error: %s\n", MsneGroupGetLastError(group)); return 1; }
    } else if (MsneGroupRender(group, sensor, info.lens, spp) != 0) { fprintf(stderr, "error: %s\n", MsneGroupGetLastError(group)); return 1; }
    logger.log("render");
    if (MsneSaveSensorExr(MsneGroupContext(group, 0), sensor, extent, out.c_str()) != 0) { fprintf(stderr, "error writing %s: %s\n", out.c_str(), MsneGetIoError()); return 1; }
    logger.log("write exr");
    MsneStats st; double gather_ms = 0.0; uint64_t gathers = 0;
    if (MsneGroupGetStats(group, &st, &gather_ms, &gathers) == 0)
        printf("%u triangles, %llu samples, %llu rays on %u GPU%s (film gather: %s, %llu x, %.3f ms)\n", info.triangles, (unsigned long long)st.samples, (unsigned long long)(st.closest_rays + st.shadow_rays),
               gpus, gpus == 1 ? "" : "s", MsneGroupTransport(group), (unsigned long long)gathers, gather_ms);
    MsneGroupDestroy(group);
    return 0;
}
