// offline.cpp — the headless renderer of the reference (offline/main.zig:27-203) on top of the C ABI only:
//     offline <in.glb> <skybox.exr> <out.exr> [spp=16] [--width W --height H --max-bounces N --env-samples N --mesh-samples N]
// Same positional arguments, extension checks, defaults (1280x720, 16 spp, max_bounces 1024, one env + one mesh light
// sample per bounce, samples_per_run 1 — offline/main.zig:41-50,106-111) and the same timing lines (:59-76).
#include "../../include/moonshine_amd.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: offline <in.glb> <skybox.exr> <out.exr> [spp] [--width W] [--height H] [--max-bounces N] [--env-samples N] [--mesh-samples N]\n"); return 2; }
    const std::string in = argv[1], sky = argv[2], out = argv[3];
    if (!has_ext(in, ".glb")) { fprintf(stderr, "error: OnlySupportsGlbInput\n"); return 2; }
    if (!has_ext(sky, ".exr")) { fprintf(stderr, "error: OnlySupportsExrSkybox\n"); return 2; }
    if (!has_ext(out, ".exr")) { fprintf(stderr, "error: OnlySupportsExrOutput\n"); return 2; }
    uint32_t spp = 16; Extent2D extent{ 1280, 720 };
    MsnePipelineOpts opts{ 1, 1024, 1, 1, 1, 1, 1 };
    int a = 4;
    if (a < argc && argv[a][0] != '-') spp = (uint32_t)atoi(argv[a++]);
    for (; a + 1 < argc; a += 2) {
        const std::string k = argv[a]; const uint32_t v = (uint32_t)atoi(argv[a + 1]);
        if (k == "--width") extent.width = v; else if (k == "--height") extent.height = v; else if (k == "--max-bounces") opts.max_bounces = v;
        else if (k == "--env-samples") opts.env_samples_per_bounce = v; else if (k == "--mesh-samples") opts.mesh_samples_per_bounce = v;
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    IntervalLogger logger;
    HdMoonshine* ctx = MsneCreate(nullptr);
    if (!ctx) { fprintf(stderr, "error: %s\n", MsneGetLastError(nullptr)); return 1; }
    logger.log("set up initial state");
    MsneGlbInfo info;
    if (MsneLoadGlb(ctx, in.c_str(), &info) != 0) { fprintf(stderr, "error loading %s: %s\n", in.c_str(), MsneGetIoError()); return 1; }
    if (MsneSetBackgroundExr(ctx, sky.c_str()) != 0) { fprintf(stderr, "error loading %s: %s\n", sky.c_str(), MsneGetIoError()); return 1; }
    const SensorHandle sensor = HdMoonshineCreateSensor(ctx, extent);
    logger.log("load world");
    if (MsneSetPipeline(ctx, &opts) != 0) { fprintf(stderr, "error: %s\n", MsneGetLastError(ctx)); return 1; }
    logger.log("create pipeline");
    if (MsneRender(ctx, sensor, info.lens, spp, 1) != 0) { fprintf(stderr, "error: %s\n", MsneGetLastError(ctx)); return 1; }
    logger.log("render");
    if (MsneSaveSensorExr(ctx, sensor, extent, out.c_str()) != 0) { fprintf(stderr, "error writing %s: %s\n", out.c_str(), MsneGetIoError()); return 1; }
    logger.log("write exr");
    MsneStats st;
    if (MsneGetStats(ctx, &st) == 0) printf("%u triangles, %llu samples, %llu rays\n", info.triangles, (unsigned long long)st.samples, (unsigned long long)(st.closest_rays + st.shadow_rays));
    HdMoonshineDestroy(ctx);
    return 0;
}
