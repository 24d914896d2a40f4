// host.h — host-side scene I/O above the C ABI (include/moonshine_amd.h): GLB import (engine/hrtsystem/World.zig:44-363,
// Camera.zig:26-51), EXR codec (engine/fileformats/exr.zig:126-231), PNG decode for glTF images (zigimg in the
// reference).  Plain C++17, no GPU code: everything here talks to the renderer only through a SceneSink, i.e. the
// public entry points, so the same importer can drive any implementation of the ABI.
#pragma once
#include "../../include/moonshine_amd.h"
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace msne_host {

struct Image { uint32_t w = 0, h = 0; std::vector<float> rgba; };            // row 0 = top
struct Image8 { uint32_t w = 0, h = 0; std::vector<uint8_t> rgb; };            // 8-bit RGB

bool read_file(const std::string& path, std::vector<uint8_t>& out);
bool exr_load(const std::string& path, Image& img, std::string& err);
bool exr_save_rgb(const std::string& path, const float* rgba, uint32_t w, uint32_t h, std::string& err);
bool png_decode(const uint8_t* data, size_t n, Image8& img, std::string& err);
// runs job(i) for i in [0, n) on the host threads this process may use (affinity mask, at most 32); job must not throw
void parallel_for(uint32_t n, const std::function<void(uint32_t)>& job);

// the subset of the C ABI the importer needs (bound to HdMoonshine*/Msne* by scene_io.cpp)
struct SceneSink {
    void* ctx;
    int64_t (*create_mesh)(void*, const F32x3*, const F32x3*, const F32x2*, size_t, size_t, const U32x3*, size_t);
    int64_t (*create_texture)(void*, const void*, Extent2D, int);
    int64_t (*solid1)(void*, float);
    int64_t (*solid2)(void*, F32x2);
    int64_t (*solid3)(void*, F32x3);
    int64_t (*create_material)(void*, const MsneMaterialDesc*);
    int64_t (*create_instance)(void*, Mat3x4, const Geometry*, size_t, bool);
    int64_t (*create_lens)(void*, Lens);
};
struct GlbSummary { uint32_t meshes = 0, materials = 0, instances = 0, textures = 0, triangles = 0; int64_t lens = -1; };
bool glb_import(const std::string& path, const SceneSink& sink, GlbSummary& out, std::string& err);

}  // namespace msne_host
