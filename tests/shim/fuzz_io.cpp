// Test-only driver: runs the host-side parsers (moonshine_amd/host/{glb,png,exr}.cpp) over the files named on the command
// line with a scene sink that only counts.  Built with -fsanitize=address,undefined by tests/test_scene_io.py and fed
// mutated files: a parser may reject a file, it may not read or write out of bounds.
#include "../../moonshine_amd/host/host.h"
#include <cstdio>
#include <cstring>
using namespace msne_host;
static int64_t n_handles = 0;
static int64_t s_mesh(void*, const F32x3* p, const F32x3* n, const F32x2* t, size_t pc, size_t ac, const U32x3* i, size_t ic) {
    // touch every element the importer claims to hand over
    double acc = 0;
    for (size_t k = 0; k < pc; k++) acc += p[k].x + p[k].y + p[k].z;
    if (n) for (size_t k = 0; k < ac; k++) acc += n[k].x;
    if (t) for (size_t k = 0; k < ac; k++) acc += t[k].x;
    for (size_t k = 0; k < ic; k++) acc += i[k].x + i[k].y + i[k].z;
    return acc == 12345.678 ? -2 : n_handles++;
}
static int64_t s_tex(void*, const void* b, Extent2D e, int f) {
    static const size_t texel[7] = { 4, 2, 1, 16, 8, 4, 8 };   // MsneTextureFormat
    if (f < 0 || f > 6) return -2;
    const size_t bytes = (size_t)e.width * e.height * texel[f];
    unsigned acc = 0;
    for (size_t k = 0; k < bytes; k++) acc += ((const uint8_t*)b)[k];
    return acc == 0xdeadbeef ? -2 : n_handles++;
}
static int64_t s_1(void*, float) { return n_handles++; }
static int64_t s_2(void*, F32x2) { return n_handles++; }
static int64_t s_3(void*, F32x3) { return n_handles++; }
static int64_t s_mat(void*, const MsneMaterialDesc*) { return n_handles++; }
static int64_t s_inst(void*, Mat3x4, const Geometry* g, size_t n, bool) { unsigned acc = 0; for (size_t k = 0; k < n; k++) acc += g[k].mesh + g[k].material; return acc == 0xdeadbeef ? -2 : n_handles++; }
static int64_t s_lens(void*, Lens) { return n_handles++; }

int main(int argc, char** argv) {
    int accepted = 0;
    for (int a = 1; a < argc; a++) {
        const char* path = argv[a]; const size_t L = strlen(path);
        std::string err;
        bool ok = false;
        if (L > 4 && !strcmp(path + L - 4, ".glb")) { SceneSink s{ nullptr, s_mesh, s_tex, s_1, s_2, s_3, s_mat, s_inst, s_lens }; GlbSummary sum; ok = glb_import(path, s, sum, err); }
        else if (L > 4 && !strcmp(path + L - 4, ".exr")) { Image img; ok = exr_load(path, img, err); if (ok && img.rgba.size() != (size_t)img.w * img.h * 4) return 3; }
        else if (L > 4 && !strcmp(path + L - 4, ".png")) { std::vector<uint8_t> d; Image8 img; if (read_file(path, d)) ok = png_decode(d.data(), d.size(), img, err); if (ok && img.rgb.size() != (size_t)img.w * img.h * 3) return 3; }
        accepted += ok;
    }
    printf("accepted %d of %d\n", accepted, argc - 1);
    return 0;
}
