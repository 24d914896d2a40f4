// Test-only glue: runs the product's host-side importers (moonshine_amd/host/*.cpp — plain C++, no GPU) against the
// CPU oracle's scene API, so a GLB / EXR pair can be loaded into the oracle exactly as MsneLoadGlb /
// MsneSetBackgroundExr load it into the HIP library.  Lives under tests/: the product never links the oracle.
#include "../../moonshine_amd/host/host.h"
#include <cstring>
struct OrcContext;
extern "C" {
int64_t OrcCreateMesh(OrcContext*, const F32x3*, const F32x3*, const F32x2*, size_t, size_t, const U32x3*, size_t);
int64_t OrcCreateTexture(OrcContext*, const void*, Extent2D, int);
int64_t OrcCreateSolidTexture1(OrcContext*, float);
int64_t OrcCreateSolidTexture2(OrcContext*, F32x2);
int64_t OrcCreateSolidTexture3(OrcContext*, F32x3);
int64_t OrcCreateMaterial(OrcContext*, const MsneMaterialDesc*);
int64_t OrcCreateInstance(OrcContext*, Mat3x4, const Geometry*, size_t, bool);
int64_t OrcCreateLens(OrcContext*, Lens);
int OrcSetBackground(OrcContext*, const float*, Extent2D);
}
using namespace msne_host;
static std::string g_err;
static int64_t s_mesh(void* c, const F32x3* p, const F32x3* n, const F32x2* t, size_t pc, size_t ac, const U32x3* i, size_t ic) { return OrcCreateMesh((OrcContext*)c, p, n, t, pc, ac, i, ic); }
static int64_t s_tex(void* c, const void* b, Extent2D e, int f) { return OrcCreateTexture((OrcContext*)c, b, e, f); }
static int64_t s_1(void* c, float v) { return OrcCreateSolidTexture1((OrcContext*)c, v); }
static int64_t s_2(void* c, F32x2 v) { return OrcCreateSolidTexture2((OrcContext*)c, v); }
static int64_t s_3(void* c, F32x3 v) { return OrcCreateSolidTexture3((OrcContext*)c, v); }
static int64_t s_mat(void* c, const MsneMaterialDesc* d) { return OrcCreateMaterial((OrcContext*)c, d); }
static int64_t s_inst(void* c, Mat3x4 t, const Geometry* g, size_t n, bool v) { return OrcCreateInstance((OrcContext*)c, t, g, n, v); }
static int64_t s_lens(void* c, Lens l) { return OrcCreateLens((OrcContext*)c, l); }
extern "C" {
const char* ShimError() { return g_err.c_str(); }
int ShimLoadGlb(OrcContext* c, const char* path, uint32_t info[6]) {
    SceneSink s{ c, s_mesh, s_tex, s_1, s_2, s_3, s_mat, s_inst, s_lens };
    GlbSummary sum;
    if (!glb_import(path, s, sum, g_err)) return -1;
    info[0] = sum.meshes; info[1] = sum.materials; info[2] = sum.instances; info[3] = sum.textures; info[4] = sum.triangles; info[5] = (uint32_t)sum.lens;
    return 0;
}
// png_decode on its own: rgb_out == NULL returns the extent only
int ShimPngDecode(const uint8_t* data, size_t n, uint8_t* rgb_out, uint32_t wh[2]) {
    Image8 img;
    if (!png_decode(data, n, img, g_err)) return -1;
    wh[0] = img.w; wh[1] = img.h;
    if (rgb_out) memcpy(rgb_out, img.rgb.data(), img.rgb.size());
    return 0;
}
int ShimSetBackgroundExr(OrcContext* c, const char* path) {
    Image img;
    if (!exr_load(path, img, g_err)) return -1;
    return OrcSetBackground(c, img.rgba.data(), Extent2D{ img.w, img.h });
}
}
