// scene_io.cpp — file-level entry points of the C ABI (Scene.fromGlbExr, engine/hrtsystem/Scene.zig:28-62, and
// Rgba2D.save, engine/fileformats/exr.zig:137-206), implemented purely on top of the public HdMoonshine*/Msne* calls.
#include "host.h"
#include <cstring>
#include <string>

using namespace msne_host;

static thread_local std::string g_io_error;

static int64_t sink_mesh(void* c, const F32x3* p, const F32x3* n, const F32x2* t, size_t pc, size_t ac, const U32x3* i, size_t ic) { return MsneCreateMesh((HdMoonshine*)c, p, n, t, pc, ac, i, ic); }
static int64_t sink_tex(void* c, const void* b, Extent2D e, int f) { return MsneCreateTexture((HdMoonshine*)c, b, e, (MsneTextureFormat)f); }
static int64_t sink_s1(void* c, float v) { return HdMoonshineCreateSolidTexture1((HdMoonshine*)c, v, ""); }
static int64_t sink_s2(void* c, F32x2 v) { return HdMoonshineCreateSolidTexture2((HdMoonshine*)c, v, ""); }
static int64_t sink_s3(void* c, F32x3 v) { return HdMoonshineCreateSolidTexture3((HdMoonshine*)c, v, ""); }
static int64_t sink_mat(void* c, const MsneMaterialDesc* d) { return MsneCreateMaterial((HdMoonshine*)c, d); }
static int64_t sink_inst(void* c, Mat3x4 t, const Geometry* g, size_t n, bool v) { return HdMoonshineCreateInstance((HdMoonshine*)c, t, g, n, v); }
static int64_t sink_lens(void* c, Lens l) { return HdMoonshineCreateLens((HdMoonshine*)c, l); }

extern "C" {

const char* MsneGetIoError(void) { return g_io_error.c_str(); }

int MsneLoadGlb(HdMoonshine* ctx, const char* path, MsneGlbInfo* info) {
    if (!ctx || !path) { g_io_error = "bad arguments"; return -1; }
    SceneSink s{ ctx, sink_mesh, sink_tex, sink_s1, sink_s2, sink_s3, sink_mat, sink_inst, sink_lens };
    GlbSummary sum;
    if (!glb_import(path, s, sum, g_io_error)) return -1;
    if (info) { info->meshes = sum.meshes; info->materials = sum.materials; info->instances = sum.instances; info->textures = sum.textures; info->triangles = sum.triangles; info->lens = (LensHandle)sum.lens; }
    return 0;
}

int MsneSetBackgroundExr(HdMoonshine* ctx, const char* path) {
    Image img;
    if (!ctx || !path || !exr_load(path, img, g_io_error)) return -1;
    if (MsneSetBackground(ctx, img.rgba.data(), Extent2D{ img.w, img.h }) != 0) { g_io_error = MsneGetLastError(ctx); return -1; }
    return 0;
}

int MsneSaveSensorExr(HdMoonshine* ctx, SensorHandle sensor, Extent2D extent, const char* path) {
    const float* px = ctx ? HdMoonshineGetSensorData(ctx, sensor) : nullptr;
    if (!px || !path) { g_io_error = "bad sensor"; return -1; }
    return exr_save_rgb(path, px, extent.width, extent.height, g_io_error) ? 0 : -1;
}

// codec access for tests and tools: load into a caller buffer (two-call pattern) / save an RGBA f32 image as B,G,R floats
int MsneExrLoad(const char* path, float* rgba_out, Extent2D* extent) {
    Image img;
    if (!path || !extent || !exr_load(path, img, g_io_error)) return -1;
    if (rgba_out && extent->width == img.w && extent->height == img.h) memcpy(rgba_out, img.rgba.data(), img.rgba.size() * 4);
    extent->width = img.w; extent->height = img.h;
    return 0;
}
int MsneExrSave(const char* path, const float* rgba, Extent2D extent) { return path && rgba && exr_save_rgb(path, rgba, extent.width, extent.height, g_io_error) ? 0 : -1; }

}  // extern "C"
