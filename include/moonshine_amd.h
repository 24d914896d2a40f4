/*
 * moonshine_amd.h — C ABI of libmoonshine_amd.so, the MI355X (gfx950) drop-in for the
 * shaders/hrtsystem + engine/hrtsystem hot path of ashpil/moonshine.
 *
 * Part 1 is, symbol for symbol and layout for layout, the reference's existing C ABI
 * (hydra/moonshine.h:10-95, implemented by hydra/hydra.zig:107-558): hydra/*.cpp links
 * against this library unchanged.  Part 2 ("Msne*") exposes the parts of the Zig API of
 * engine/hrtsystem that the reference's `offline`, `online` and engine/tests.zig call
 * directly and that moonshine.h never exported (material variants, backgrounds, pipeline
 * specialization constants, batched launches, sharded films); each entry cites the
 * reference interface it stands in for.
 *
 * Conventions (README.md:56-58): +z up, Mat3x4 = 3 rows of float4, row-major object->world.
 * All handles are dense uint32_t indices.  Calls on one context are serialized by one
 * mutex (hydra.zig:76-78).  Nothing throws across the ABI; functions that can fail return
 * false / NULL / negative codes.  Plain pointers and sizes only — no torch, no HIP types.
 */
#ifndef MOONSHINE_AMD_H
#define MOONSHINE_AMD_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Part 1 — hydra/moonshine.h, verbatim layouts                        */
/* ------------------------------------------------------------------ */
typedef uint32_t MeshHandle;      /* moonshine.h:10 */
typedef uint32_t ImageHandle;     /* :11 */
typedef uint32_t MaterialHandle;  /* :12 */
typedef uint32_t SensorHandle;    /* :13 */
typedef uint32_t LensHandle;      /* :14 */
typedef uint32_t InstanceHandle;  /* :15 */

typedef struct F32x2 { float x, y; } F32x2;                 /* :17 */
typedef struct F32x3 { float x, y, z; } F32x3;              /* :21 */
typedef struct F32x4 { float x, y, z, w; } F32x4;           /* :25 */
typedef struct U32x3 { uint32_t x, y, z; } U32x3;           /* :29 */
typedef struct Mat3x4 { F32x4 x, y, z; } Mat3x4;            /* :33 */
typedef struct Geometry {                                   /* :37  (Accel.zig:40-44) */
    MeshHandle mesh;
    MaterialHandle material;
    bool sampled;
} Geometry;
typedef struct Extent2D { uint32_t width, height; } Extent2D; /* :43 */
typedef struct Lens {                                       /* :48  (Camera.zig:18-25) */
    F32x3 origin;
    F32x3 forward;
    F32x3 up;
    float vfov;          /* radians */
    float aperture;
    float focus_distance;
} Lens;
typedef struct Material {                                   /* :57  (hydra.zig:38-42) StandardPBR only */
    ImageHandle normal;
    ImageHandle emissive;
    ImageHandle color;
    ImageHandle metalness;
    ImageHandle roughness;
    float ior;
} Material;
typedef enum TextureFormat { f16x4, u8x4_srgb } TextureFormat; /* :66 */

typedef struct HdMoonshine HdMoonshine;                      /* :71 */

/* :72  hydra.zig:107-143.  Device 0 (or $MSNE_DEVICE); pipeline constants {1,1024,0,0,false,false,false}
 * (hydra.zig:97-105); 1x1 white default background (BackgroundManager.zig:116).  NULL on failure. */
HdMoonshine* HdMoonshineCreate(void);
void HdMoonshineDestroy(HdMoonshine*);                                          /* :73  hydra.zig:542 */
/* :74  hydra.zig:145-363.  Synchronous: flush deferred material edits, rebuild the acceleration
 * structure if instances changed, one launch of samples_per_run samples per pixel, copy the film
 * into the sensor's host buffer, sample_count += samples_per_run. */
bool HdMoonshineRender(HdMoonshine*, SensorHandle, LensHandle);
bool HdMoonshineRebuildPipeline(HdMoonshine*);                                  /* :75  hydra.zig:365 — clears all sensors */
/* :76  hydra.zig:374-384.  normals/texcoords are face-varying (index_count*3 entries) or NULL. Data is copied. */
MeshHandle HdMoonshineCreateMesh(HdMoonshine*, const F32x3* positions, const F32x3* normals, const F32x2* texcoords,
                                 size_t position_count, const U32x3* indices, size_t index_count);
ImageHandle HdMoonshineCreateSolidTexture1(HdMoonshine*, float, const char* name);     /* :77 */
ImageHandle HdMoonshineCreateSolidTexture2(HdMoonshine*, F32x2, const char* name);     /* :78 */
ImageHandle HdMoonshineCreateSolidTexture3(HdMoonshine*, F32x3, const char* name);     /* :79 */
ImageHandle HdMoonshineCreateRawTexture(HdMoonshine*, uint8_t* data, Extent2D, TextureFormat, const char* name); /* :80 */
MaterialHandle HdMoonshineCreateMaterial(HdMoonshine*, Material);                      /* :81 */
void HdMoonshineSetMaterialNormal(HdMoonshine*, MaterialHandle, ImageHandle);          /* :82  deferred to next Render */
void HdMoonshineSetMaterialEmissive(HdMoonshine*, MaterialHandle, ImageHandle);        /* :83 */
void HdMoonshineSetMaterialColor(HdMoonshine*, MaterialHandle, ImageHandle);           /* :84 */
void HdMoonshineSetMaterialMetalness(HdMoonshine*, MaterialHandle, ImageHandle);       /* :85 */
void HdMoonshineSetMaterialRoughness(HdMoonshine*, MaterialHandle, ImageHandle);       /* :86 */
void HdMoonshineSetMaterialIOR(HdMoonshine*, MaterialHandle, float);                   /* :87 */
/* :88  hydra.zig:483-493 → Accel.uploadInstance (Accel.zig:189); clears all sensors */
InstanceHandle HdMoonshineCreateInstance(HdMoonshine*, Mat3x4, const Geometry*, size_t geometry_count, bool visible);
void HdMoonshineDestroyInstance(HdMoonshine*, InstanceHandle);                         /* :89  = hide (hydra.zig:495) */
void HdMoonshineSetInstanceTransform(HdMoonshine*, InstanceHandle, Mat3x4);            /* :90 */
void HdMoonshineSetInstanceVisibility(HdMoonshine*, InstanceHandle, bool);             /* :91 */
SensorHandle HdMoonshineCreateSensor(HdMoonshine*, Extent2D);                          /* :92 */
/* :93  float4[w*h], row-major, library-owned pinned host memory, valid until Destroy; alpha = launches */
float* HdMoonshineGetSensorData(const HdMoonshine*, SensorHandle);
LensHandle HdMoonshineCreateLens(HdMoonshine*, Lens);                                  /* :94 */
void HdMoonshineSetLens(HdMoonshine*, LensHandle, Lens);                               /* :95  clears all sensors */

/* ------------------------------------------------------------------ */
/* Part 2 — engine/hrtsystem Zig API that moonshine.h does not export  */
/* ------------------------------------------------------------------ */

/* MaterialManager.zig:45-50 ≡ shaders/hrtsystem/world.hlsl:31-36 */
typedef enum MsneMaterialType {
    MSNE_MATERIAL_GLASS = 0,
    MSNE_MATERIAL_LAMBERT = 1,
    MSNE_MATERIAL_PERFECT_MIRROR = 2,
    MSNE_MATERIAL_STANDARD_PBR = 3,
} MsneMaterialType;

/* MaterialManager.MaterialInfo (MaterialManager.zig:22-77): normal + emissive + tagged variant.
 * glass uses {ior}; lambert {color}; perfect_mirror nothing; standard_pbr {color,metalness,roughness,ior}. */
typedef struct MsneMaterialDesc {
    ImageHandle normal;
    ImageHandle emissive;
    uint32_t type;        /* MsneMaterialType */
    ImageHandle color;
    ImageHandle metalness;
    ImageHandle roughness;
    float ior;
} MsneMaterialDesc;

/* vk.Format values the Zig API is called with (World.zig:72,101,146,184,196; MaterialManager.zig:364-388; hydra.zig:47-52) */
typedef enum MsneTextureFormat {
    MSNE_FORMAT_R8G8B8A8_SRGB = 0,
    MSNE_FORMAT_R8G8_UNORM = 1,
    MSNE_FORMAT_R8_UNORM = 2,
    MSNE_FORMAT_R32G32B32A32_SFLOAT = 3,
    MSNE_FORMAT_R32G32_SFLOAT = 4,
    MSNE_FORMAT_R32_SFLOAT = 5,
    MSNE_FORMAT_R16G16B16A16_SFLOAT = 6,
} MsneTextureFormat;

/* StandardPipeline.SpecConstants (pipeline.zig:319-327 ↔ shaders/hrtsystem/main.hlsl:34-40).
 * Reference defaults {1,4,1,1,true,true,true}; offline uses max_bounces=1024 (offline/main.zig:106-111). */
typedef struct MsnePipelineOpts {
    uint32_t samples_per_run;
    uint32_t max_bounces;
    uint32_t env_samples_per_bounce;
    uint32_t mesh_samples_per_bounce;
    uint32_t flip_image;                    /* bool32 */
    uint32_t indexed_attributes;            /* bool32 */
    uint32_t two_component_normal_texture;  /* bool32 */
} MsnePipelineOpts;

/* Context creation with explicit placement.  The image is cut into tile_size² tiles; tile t belongs to
 * shard (t mod shard_count); this context renders only the tiles of `shard_index`.
 * shard_count=1 renders everything (the reference's single-device behaviour, VulkanContext.zig:313-326). */
#define MSNE_DEFAULT_TILE_SIZE 16u   /* one k_shade workgroup per tile; rank load spread at 8 shards 2.5 % (64x64 tiles: 9.6 %) */
typedef struct MsneConfig {
    int32_t device;        /* HIP device ordinal; -1 = $MSNE_DEVICE or 0 */
    uint32_t tile_size;    /* 0 = MSNE_DEFAULT_TILE_SIZE */
    uint32_t shard_index;
    uint32_t shard_count;  /* 0 = 1 */
} MsneConfig;

typedef struct MsneStats {
    uint64_t closest_rays;   /* Intersection::find equivalents traced (intersection.hlsl:18-22) */
    uint64_t shadow_rays;    /* ShadowIntersection::hit equivalents traced (intersection.hlsl:33-46) */
    uint64_t samples;        /* camera paths started (main.hlsl:83-92) */
    uint64_t launches;       /* recordTraceRays equivalents (pipeline.zig:269-271) */
    double   trace_closest_ms; /* HIP-event time inside k_trace_closest since the last reset (on the render stream) */
    double   trace_shadow_ms;
    double   shade_ms;
    double   render_ms;        /* first raygen launch → film complete, summed over MsneRender calls */
    uint64_t trace_closest_launches;
    uint64_t trace_shadow_launches;
    uint64_t shade_launches;
} MsneStats;

HdMoonshine* MsneCreate(const MsneConfig*);                       /* HdMoonshineCreate with placement */

/* MeshManager.upload (MeshManager.zig:70-156).  attribute_count = length of normals/texcoords:
 * position_count when the pipeline has indexed_attributes=true (glTF mode, world.hlsl:130),
 * index_count*3 when false (Hydra mode).  Indices are range-checked here; a render whose pipeline would read past
 * attribute_count on a mesh some instance uses fails with an error instead of reading out of bounds.  Negative on error. */
int64_t MsneCreateMesh(HdMoonshine*, const F32x3* positions, const F32x3* normals, const F32x2* texcoords,
                       size_t position_count, size_t attribute_count, const U32x3* indices, size_t index_count);
/* TextureManager.upload with a .raw source (MaterialManager.zig:351-445) */
int64_t MsneCreateTexture(HdMoonshine*, const void* bytes, Extent2D, MsneTextureFormat);
/* MaterialManager.upload (MaterialManager.zig:140-170) with any variant */
int64_t MsneCreateMaterial(HdMoonshine*, const MsneMaterialDesc*);
/* Accel.recordUpdateSingleMaterial (Accel.zig:609-628; caller online/main.zig:229-233): geometry `geometry_index` of `instance` uses `material` from the next
 * render on.  One record update — no rebuild; nothing is cleared (the reference's caller clears the sensor itself).  0 on success, negative for unknown handles. */
int MsneSetGeometryMaterial(HdMoonshine*, InstanceHandle instance, uint32_t geometry_index, MaterialHandle material);
/* StandardPipeline.create / recreate (pipeline.zig:85,180): change specialization constants; clears all sensors */
int MsneSetPipeline(HdMoonshine*, const MsnePipelineOpts*);
int MsneGetPipeline(const HdMoonshine*, MsnePipelineOpts*);
/* BackgroundManager.addBackground (BackgroundManager.zig:142-394): equirectangular RGBA32F, row 0 = top (theta=0).
 * Runs the three shaders/background kernels' equivalents on the GPU; clears all sensors. */
int MsneSetBackground(HdMoonshine*, const float* rgba, Extent2D);
/* `launches` back-to-back HdMoonshineRender-equivalents without host round trips (offline/main.zig:131-165 spp loop);
 * one film readback at the end when `readback` is nonzero.  0 on success. */
int MsneRender(HdMoonshine*, SensorHandle, LensHandle, uint32_t launches, int readback);
/* Pre-allocates the wavefront state for MsneRender calls of up to `launches` launches on this sensor (otherwise it is
 * allocated on first use, inside that call).  Up to $MSNE_MAX_INFLIGHT (default 160 Mi) paths are traced concurrently. */
int MsneReserve(HdMoonshine*, SensorHandle, uint32_t launches);
/* The in-flight budget itself, in paths (288 B of wavefront state each at one env + one mesh light sample).  Contexts that share one GPU
 * share its memory: MsneGroupCreate divides the default among the members it places on the same device. */
int MsneSetMaxInflight(HdMoonshine*, uint64_t paths);
uint64_t MsneGetMaxInflight(const HdMoonshine*);
void MsneClearSensor(HdMoonshine*, SensorHandle);                 /* Sensor.clear (core/Sensor.zig:81-83) */
uint32_t MsneGetSampleCount(const HdMoonshine*, SensorHandle);    /* Sensor.sample_count (core/Sensor.zig:12) */

/* Sharded film access for multi-GPU gathers (no reference equivalent: the reference is single-device).
 * The packed film holds this shard's tiles in tile order, each tile tile_size² float4, row-major inside the tile. */
uint64_t MsneGetShardTileCount(const HdMoonshine*, SensorHandle);
void* MsneGetPackedFilmDevicePtr(const HdMoonshine*, SensorHandle);            /* device pointer, float4[tiles*ts*ts] */
/* root side: scatter `shard_count` packed films (concatenated in shard order, each padded to max tiles per shard)
 * from device memory into the sensor's full row-major film and its host buffer.  The gathered buffer must be complete
 * when this is called (the library's streams are non-blocking: synchronise the stream that filled it first). */
int MsneUnpackGatheredFilm(HdMoonshine*, SensorHandle, const void* gathered_device_ptr, uint32_t shard_count);
uint64_t MsneGetPackedFilmStride(const HdMoonshine*, SensorHandle);   /* float4 per shard in a gathered buffer: the largest shard's tiles * tile_size^2 */


/* ObjectPicker.getClickedObject (ObjectPicker.zig:89-128) = the raygen of shaders/hrtsystem/input.hlsl:24-69: ONE closest-hit
 * ray through `normalized_coords` (0..1 across the sensor, y down: uv = (x, 1 - y)), lens sample (0,0), lens used as given
 * (the shader zeroes the aperture of a copy it then does not use).  instance_index = -1 on a miss. */
typedef struct MsneClickData { int32_t instance_index; uint32_t geometry_index, primitive_index; F32x2 barycentrics; } MsneClickData;
int MsnePick(HdMoonshine*, SensorHandle, LensHandle, F32x2 normalized_coords, MsneClickData* out);

int MsneGetStats(const HdMoonshine*, MsneStats*);
void MsneResetStats(HdMoonshine*);
const char* MsneGetLastError(const HdMoonshine*);   /* NULL ctx → last creation error */

/* ---- file-level entry points: Scene.fromGlbExr (Scene.zig:28-62), Rgba2D.load/save (exr.zig:137-229) ---- */
typedef struct MsneGlbInfo { uint32_t meshes, materials, instances, textures, triangles; LensHandle lens; } MsneGlbInfo;
/* World.fromGlb + Camera.Lens.fromGlb (World.zig:233-363, Camera.zig:26-51): appends the GLB's meshes, materials,
 * instances and its first camera (as a lens) to the context.  0 on success; MsneGetIoError() says why not. */
int MsneLoadGlb(HdMoonshine*, const char* glb_path, MsneGlbInfo* info_out);
/* Rgba2D.load + BackgroundManager.addBackground (Scene.zig:49-55): equirectangular EXR → environment */
int MsneSetBackgroundExr(HdMoonshine*, const char* exr_path);
/* Rgba2D.save (exr.zig:137-206): the sensor's host buffer as a 3-channel (B,G,R) FLOAT scanline EXR, ZIP blocks of 16 lines (tinyexr's header defaults); alpha is dropped */
int MsneSaveSensorExr(HdMoonshine*, SensorHandle, Extent2D, const char* exr_path);
/* the EXR codec itself.  Load: call with rgba_out == NULL to get the extent, then again with a buffer of w*h*4 floats. */
int MsneExrLoad(const char* exr_path, float* rgba_out, Extent2D* extent_inout);
int MsneExrSave(const char* exr_path, const float* rgba, Extent2D extent);
const char* MsneGetIoError(void);

/* ---- native multi-GPU rendering: one context per GPU in ONE process (no reference equivalent: VulkanContext.zig:313-326 picks one
 * device).  Image tiles shard over the members (tile t -> member t mod n), every member holds the whole scene, the only collective
 * in the data path is ONE ncclGather (RCCL over xGMI) of the packed films to member 0, followed by k_unpack_film there.  Members on
 * the same GPU (tests, one-GPU machines) gather with a device copy instead.  Scene calls go to every member — directly through
 * MsneGroupContext(g, i), or through the replicating wrappers below, which run the members' calls on one host thread each. ---- */
typedef struct MsneGroup MsneGroup;
MsneGroup* MsneGroupCreate(const int32_t* devices /* HIP ordinals, NULL = 0..n-1 modulo the device count */, uint32_t n, uint32_t tile_size /* 0 = default */);
void MsneGroupDestroy(MsneGroup*);
uint32_t MsneGroupSize(const MsneGroup*);
HdMoonshine* MsneGroupContext(MsneGroup*, uint32_t member);
int MsneGroupLoadGlb(MsneGroup*, const char* glb_path, MsneGlbInfo* info_out);              /* Scene.fromGlbExr on every member (Scene.zig:28-62) */
int MsneGroupSetBackgroundExr(MsneGroup*, const char* exr_path);
int MsneGroupSetPipeline(MsneGroup*, const MsnePipelineOpts*);
int64_t MsneGroupCreateSensor(MsneGroup*, Extent2D);
/* offline/main.zig:131-203: `launches` launches on every member's tiles, concurrently; gather; unpack.  The full image is then in
 * HdMoonshineGetSensorData(MsneGroupContext(g, 0), sensor). */
int MsneGroupRender(MsneGroup*, SensorHandle, LensHandle, uint32_t launches);
/* The frame loop of `online` without a window (online/main.zig:287-305,415-416): per frame, clear the sensor when it is past
 * max_sample_count, launch samples_per_run samples while it is below (always when max_sample_count == 0), present, count.  Presenting
 * = gather + unpack + `present(user, frame, rgba float4[w*h], sample_count)`, every `gather_every` frames and after the last one. */
typedef int (*MsnePresentFn)(void* user, uint32_t frame, const float* rgba, uint32_t sample_count);
int MsneGroupRenderProgressive(MsneGroup*, SensorHandle, LensHandle, uint32_t frames, uint32_t max_sample_count, uint32_t gather_every, MsnePresentFn present, void* user);
int MsneGroupGetStats(MsneGroup*, MsneStats* summed_out, double* gather_ms_out, uint64_t* gathers_out);   /* render_ms = the slowest member's */
const char* MsneGroupTransport(const MsneGroup*);        /* "rccl" | "copy" | "none": what the last gather used */
const char* MsneGroupGetLastError(const MsneGroup*);     /* NULL group -> last creation error */

/* ------------------------------------------------------------------ */
/* Part 3 — diagnostics for parity tests and profiling (no reference   */
/* equivalent; never needed by a renderer front end)                   */
/* ------------------------------------------------------------------ */
/* kernel_events: 1 = bracket every trace/shade launch with HIP events on the stream it is launched on (MsneStats *_ms fields);
 * 2 = the same with the kernels in stream order (the any-hit kernel of a bounce is not overlapped with the closest-hit kernel of
 * the next one), so that per-kernel durations are exclusive — what bench.py's roofline attribution pass uses;
 * traversal_counters: count BVH node visits / triangle tests inside the trace kernels. */
void MsneSetProfiling(HdMoonshine*, int kernel_events, int traversal_counters);
/* The shader clock right now, in GHz (negative on failure): one probe wave on a stream of its own reads the chip's clock counter and its 100 MHz reference counter
 * around a ~0.2 ms spin — callable from another host thread while a render is running (bench.py --sustain-seconds). */
double MsneProbeClockGhz(int device);
/* out[0] = acceleration-structure rebuilds so far, out[1] = in-place TLAS updates (instance transform edits that left the structure alone, Accel.zig:567-601) */
void MsneGetAccelStats(HdMoonshine*, uint64_t out[2]);
/* bytes of texture data the context keeps in HBM: every texture in the format it was created with (MaterialManager.zig:351-390), each rounded up to 16 B */
uint64_t MsneGetTexelPoolBytes(HdMoonshine*);
/* How acceleration structures are built from the next (re)build on.  The reference asks the driver for prefer_fast_trace builds everywhere (Accel.zig:112,259,445,645)
   and so does the default (1): one surface-area sweep over every primitive.  0 = prefer a fast BUILD (agglomerative clustering only, no sweep: about a third of the
   build time, ~4 % fewer rays per second) — for editing sessions that rebuild the TLAS every frame.  Results do not depend on it. */
void MsneSetBuildQuality(HdMoonshine*, int prefer_fast_trace);
int MsneGetTraversalCounters(HdMoonshine*, uint64_t out[20]); /* [0..3] closest {nodes,tris}, shadow {nodes,tris}; [4..19] wave-cycle profiles */
/* lane use of the traversal loop with the counters on, summed over wave iterations: out[0..11] closest-hit kernel, out[12..23] any-hit kernel —
   {iterations, lanes with a ray, lanes in the node / triangle / space body, lanes waiting for the space body, lanes with a ray that ran no body,
   iterations that ran the node / triangle / space body, lanes waiting for their triangle queue, unused} */
int MsneGetTraversalLaneUse(HdMoonshine*, uint64_t out[24]);
/* queue lengths of the last batch, per bounce b: out[4b..4b+3] = {path-queue entries, of which entries without a ray, shadow-queue entries, shadow rays traced}; returns the number of bounces written */
int MsneGetBounceCounters(HdMoonshine*, uint32_t* out, uint32_t max_bounces);
/* the last render made with kernel events on: its k_trace_closest (kind 0) / k_trace_shadow (1) / k_shade (2) launches in issue order — closest(b), shade(b), shadow(b)
   for b = 0, 1, ... — and each one's duration in ms; returns the number written */
int MsneGetLaunchTimes(HdMoonshine*, int32_t* kinds, float* ms, uint32_t max_launches);
/* rays: 7 floats each (origin, direction, tmax); out_ids: 4 per ray {hit, instance, geometry, primitive}; out_tuv: 3 per ray */
int MsneTraceRays(HdMoonshine*, const float* rays, uint32_t n, int any_hit, uint32_t* out_ids, float* out_tuv);
/* Batch probe of the device-side shading functions (the material.hlsl / light.hlsl / mappings.hlsl / math.hlsl restatements that
 * k_shade runs), one record per thread: fn 0 BSDF pdf/eval/sample, 1-3 EnvMap sample/eval/incomingRadiance on the context's
 * environment, 4-8 sampling warps, 9 Fresnel::dielectric, 10 offsetAlongNormal, 11 coordinateSystem, 12 areaMeasureToSolidAngleMeasure,
 * 13 GGX D/Lambda/G, 14 refractDir, 15 powerHeuristic, 16 Frame, 17 dTextures[i].SampleLevel, 18 MeshAttributes::lookupAndInterpolate + inWorld
 * on explicit vertex data (world.hlsl:86-176), 19 getTextureFrame on a sampled normal texel (material.hlsl:489-517), 20 Camera::generateRay from a
 * lens + extent (camera.hlsl:14-42; the host half is the make_camera() MsneRender runs).  Record widths: tests/second_source.py PROBES. */
int MsneShadeProbe(HdMoonshine*, int fn, const float* in, uint32_t n, float* out);
uint32_t MsneGetEnvSize(const HdMoonshine*);
int MsneReadEnv(HdMoonshine*, float* rgb_out, float* lum_pyramid_out);
uint32_t MsneGetAliasTable(HdMoonshine*, void* out_entries_20B, uint32_t max_entries);
int MsneReadBvh(HdMoonshine*, void* nodes_out, uint32_t* node_count, void* tris_out, uint32_t* tri_count,
                uint32_t* tlas_root, uint32_t* tlas_items_out, uint32_t* tlas_item_count);

#ifdef __cplusplus
}
#endif
#endif
