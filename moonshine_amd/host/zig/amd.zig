//! amd.zig — Zig binding of libmoonshine_amd.so (include/moonshine_amd.h), the MI355X drop-in for the reference's
//! engine/hrtsystem + shaders/hrtsystem.  This is the file INTEGRATION.md §2 tells a maintainer to add as
//! engine/hrtsystem/amd.zig; it depends on `std` only, so `offline.zig` and `furnace_test.zig` next to it build on their own:
//!
//!   zig build-exe offline.zig -lc -L../.. -lmoonshine_amd -rpath ../..
//!   zig test furnace_test.zig -lc -L../.. -lmoonshine_amd -rpath ../..
//!
//! NOT COMPILED in the image this was written in (it has no Zig toolchain — `__graft_entry__.build()` says so and skips it); the
//! declarations are checked against the header by tests/test_abi.py::test_zig_binding_declares_the_header's_entry_points
//! (names and argument counts), the layouts against the C structs the Python binding uses.
//! Every declaration cites the reference interface it stands in for.
const std = @import("std");

// vector.zig:5-243 — extern structs with the layouts of hydra/moonshine.h:17-36
pub const F32x2 = extern struct { x: f32, y: f32, pub fn new(x: f32, y: f32) F32x2 { return .{ .x = x, .y = y }; } };
pub const F32x3 = extern struct {
    x: f32, y: f32, z: f32,
    pub fn new(x: f32, y: f32, z: f32) F32x3 { return .{ .x = x, .y = y, .z = z }; }
    pub fn add(a: F32x3, b: F32x3) F32x3 { return new(a.x + b.x, a.y + b.y, a.z + b.z); }
    pub fn div_scalar(a: F32x3, s: f32) F32x3 { return new(a.x / s, a.y / s, a.z / s); }
    pub fn unit(a: F32x3) F32x3 { return a.div_scalar(@sqrt(a.x * a.x + a.y * a.y + a.z * a.z)); }
};
pub const F32x4 = extern struct { x: f32, y: f32, z: f32, w: f32 };
pub const U32x3 = extern struct { x: u32, y: u32, z: u32, pub fn new(x: u32, y: u32, z: u32) U32x3 { return .{ .x = x, .y = y, .z = z }; } };
pub const Mat3x4 = extern struct {                                   // vector.zig:245 — three rows, row-major object -> world
    x: F32x4, y: F32x4, z: F32x4,
    pub const identity = Mat3x4{ .x = .{ .x = 1, .y = 0, .z = 0, .w = 0 }, .y = .{ .x = 0, .y = 1, .z = 0, .w = 0 }, .z = .{ .x = 0, .y = 0, .z = 1, .w = 0 } };
};

pub const Ctx = opaque {};                                           // HdMoonshine (moonshine.h:71)
pub const Group = opaque {};                                         // MsneGroup: one context per GPU (no reference equivalent)
pub const Extent2D = extern struct { width: u32, height: u32 };      // vk.Extent2D
pub const Geometry = extern struct { mesh: u32, material: u32, sampled: bool };                       // Accel.zig:40-44
pub const Lens = extern struct { origin: F32x3, forward: F32x3, up: F32x3, vfov: f32, aperture: f32, focus_distance: f32 };   // Camera.zig:18-25
pub const HydraMaterial = extern struct { normal: u32, emissive: u32, color: u32, metalness: u32, roughness: u32, ior: f32 };  // hydra.zig:38-42
pub const HydraTextureFormat = enum(c_int) { f16x4, u8x4_srgb };                                      // moonshine.h:66
pub const MaterialType = enum(u32) { glass, lambert, perfect_mirror, standard_pbr };                  // MaterialManager.zig:45-50
pub const MaterialDesc = extern struct { normal: u32, emissive: u32, type: MaterialType, color: u32 = 0, metalness: u32 = 0, roughness: u32 = 0, ior: f32 = 1.5 };
pub const TextureFormat = enum(c_int) { r8g8b8a8_srgb, r8g8_unorm, r8_unorm, r32g32b32a32_sfloat, r32g32_sfloat, r32_sfloat, r16g16b16a16_sfloat };
pub const PipelineOpts = extern struct {                                                              // pipeline.zig:319-327
    samples_per_run: u32 = 1, max_bounces: u32 = 4, env_samples_per_bounce: u32 = 1, mesh_samples_per_bounce: u32 = 1,
    flip_image: u32 = 1, indexed_attributes: u32 = 1, two_component_normal_texture: u32 = 1,
};
pub const Config = extern struct { device: i32 = -1, tile_size: u32 = 0, shard_index: u32 = 0, shard_count: u32 = 0 };
pub const Stats = extern struct {
    closest_rays: u64, shadow_rays: u64, samples: u64, launches: u64,
    trace_closest_ms: f64, trace_shadow_ms: f64, shade_ms: f64, render_ms: f64,
    trace_closest_launches: u64, trace_shadow_launches: u64, shade_launches: u64,
};
pub const ClickData = extern struct { instance_index: i32, geometry_index: u32, primitive_index: u32, barycentrics: F32x2 };   // input.hlsl:24-29
pub const GlbInfo = extern struct { meshes: u32, materials: u32, instances: u32, textures: u32, triangles: u32, lens: u32 };
pub const PresentFn = *const fn (user: ?*anyopaque, frame: u32, rgba: [*]const f32, sample_count: u32) callconv(.C) c_int;

// ---- part 1: hydra/moonshine.h:72-95 (hydra.zig:107-558) ----
pub extern fn HdMoonshineCreate() ?*Ctx;
pub extern fn HdMoonshineDestroy(ctx: *Ctx) void;
pub extern fn HdMoonshineRender(ctx: *Ctx, sensor: u32, lens: u32) bool;
pub extern fn HdMoonshineRebuildPipeline(ctx: *Ctx) bool;
pub extern fn HdMoonshineCreateMesh(ctx: *Ctx, positions: [*]const F32x3, normals: ?[*]const F32x3, texcoords: ?[*]const F32x2, position_count: usize, indices: [*]const U32x3, index_count: usize) u32;
pub extern fn HdMoonshineCreateSolidTexture1(ctx: *Ctx, v: f32, name: [*:0]const u8) u32;
pub extern fn HdMoonshineCreateSolidTexture2(ctx: *Ctx, v: F32x2, name: [*:0]const u8) u32;
pub extern fn HdMoonshineCreateSolidTexture3(ctx: *Ctx, v: F32x3, name: [*:0]const u8) u32;
pub extern fn HdMoonshineCreateRawTexture(ctx: *Ctx, data: [*]u8, extent: Extent2D, format: HydraTextureFormat, name: [*:0]const u8) u32;
pub extern fn HdMoonshineCreateMaterial(ctx: *Ctx, material: HydraMaterial) u32;
pub extern fn HdMoonshineSetMaterialNormal(ctx: *Ctx, material: u32, image: u32) void;
pub extern fn HdMoonshineSetMaterialEmissive(ctx: *Ctx, material: u32, image: u32) void;
pub extern fn HdMoonshineSetMaterialColor(ctx: *Ctx, material: u32, image: u32) void;
pub extern fn HdMoonshineSetMaterialMetalness(ctx: *Ctx, material: u32, image: u32) void;
pub extern fn HdMoonshineSetMaterialRoughness(ctx: *Ctx, material: u32, image: u32) void;
pub extern fn HdMoonshineSetMaterialIOR(ctx: *Ctx, material: u32, ior: f32) void;
pub extern fn HdMoonshineCreateInstance(ctx: *Ctx, transform: Mat3x4, geometries: [*]const Geometry, count: usize, visible: bool) u32;   // Accel.uploadInstance :189
pub extern fn HdMoonshineDestroyInstance(ctx: *Ctx, instance: u32) void;
pub extern fn HdMoonshineSetInstanceTransform(ctx: *Ctx, instance: u32, transform: Mat3x4) void;       // Accel.recordUpdateSingleTransform :567
pub extern fn HdMoonshineSetInstanceVisibility(ctx: *Ctx, instance: u32, visible: bool) void;          // Accel.updateVisibility :603
pub extern fn HdMoonshineCreateSensor(ctx: *Ctx, extent: Extent2D) u32;                                // Camera.appendSensor :60
pub extern fn HdMoonshineGetSensorData(ctx: *const Ctx, sensor: u32) [*][4]f32;
pub extern fn HdMoonshineCreateLens(ctx: *Ctx, lens: Lens) u32;                                        // Camera.appendLens :69
pub extern fn HdMoonshineSetLens(ctx: *Ctx, handle: u32, lens: Lens) void;

// ---- part 2: the engine/hrtsystem calls of offline / online / tests.zig that moonshine.h never exported ----
pub extern fn MsneCreate(cfg: ?*const Config) ?*Ctx;                                                   // VulkanContext.create + World.createEmpty
pub extern fn MsneCreateMesh(ctx: *Ctx, positions: [*]const F32x3, normals: ?[*]const F32x3, texcoords: ?[*]const F32x2,
    position_count: usize, attribute_count: usize, indices: [*]const U32x3, index_count: usize) i64;  // MeshManager.upload :70
pub extern fn MsneCreateTexture(ctx: *Ctx, bytes: *const anyopaque, extent: Extent2D, format: TextureFormat) i64;   // TextureManager.upload :351
pub extern fn MsneCreateMaterial(ctx: *Ctx, desc: *const MaterialDesc) i64;                            // MaterialManager.upload :140
pub extern fn MsneSetGeometryMaterial(ctx: *Ctx, instance: u32, geometry_index: u32, material: u32) c_int;   // Accel.recordUpdateSingleMaterial :609
pub extern fn MsneSetPipeline(ctx: *Ctx, opts: *const PipelineOpts) c_int;                             // StandardPipeline.create / recreate :85,:180
pub extern fn MsneGetPipeline(ctx: *const Ctx, opts: *PipelineOpts) c_int;
pub extern fn MsneSetBackground(ctx: *Ctx, rgba: [*]const f32, extent: Extent2D) c_int;                // BackgroundManager.addBackground :142
pub extern fn MsneRender(ctx: *Ctx, sensor: u32, lens: u32, launches: u32, readback: c_int) c_int;     // recordTraceRays x launches + copy (offline/main.zig:131-203)
pub extern fn MsneReserve(ctx: *Ctx, sensor: u32, launches: u32) c_int;
pub extern fn MsneSetMaxInflight(ctx: *Ctx, paths: u64) c_int;
pub extern fn MsneGetMaxInflight(ctx: *const Ctx) u64;
pub extern fn MsneClearSensor(ctx: *Ctx, sensor: u32) void;                                            // Sensor.clear (core/Sensor.zig:81)
pub extern fn MsneGetSampleCount(ctx: *const Ctx, sensor: u32) u32;                                    // Sensor.sample_count
pub extern fn MsnePick(ctx: *Ctx, sensor: u32, lens: u32, normalized_coords: F32x2, out: *ClickData) c_int;   // ObjectPicker.getClickedObject :89
pub extern fn MsneGetStats(ctx: *const Ctx, stats: *Stats) c_int;
pub extern fn MsneResetStats(ctx: *Ctx) void;
pub extern fn MsneGetLastError(ctx: ?*const Ctx) [*:0]const u8;
// file level: Scene.fromGlbExr (Scene.zig:28-62), Rgba2D.load / save (exr.zig:137-229)
pub extern fn MsneLoadGlb(ctx: *Ctx, glb_path: [*:0]const u8, info: ?*GlbInfo) c_int;
pub extern fn MsneSetBackgroundExr(ctx: *Ctx, exr_path: [*:0]const u8) c_int;
pub extern fn MsneSaveSensorExr(ctx: *Ctx, sensor: u32, extent: Extent2D, exr_path: [*:0]const u8) c_int;
pub extern fn MsneExrLoad(exr_path: [*:0]const u8, rgba_out: ?[*]f32, extent_inout: *Extent2D) c_int;
pub extern fn MsneExrSave(exr_path: [*:0]const u8, rgba: [*]const f32, extent: Extent2D) c_int;
pub extern fn MsneGetIoError() [*:0]const u8;
// sharded films and the one-process multi-GPU group
pub extern fn MsneGetShardTileCount(ctx: *const Ctx, sensor: u32) u64;
pub extern fn MsneGetPackedFilmDevicePtr(ctx: *const Ctx, sensor: u32) ?*anyopaque;
pub extern fn MsneGetPackedFilmStride(ctx: *const Ctx, sensor: u32) u64;
pub extern fn MsneUnpackGatheredFilm(ctx: *Ctx, sensor: u32, gathered_device_ptr: *const anyopaque, shard_count: u32) c_int;
pub extern fn MsneGroupCreate(devices: ?[*]const i32, n: u32, tile_size: u32) ?*Group;
pub extern fn MsneGroupDestroy(group: *Group) void;
pub extern fn MsneGroupSize(group: *const Group) u32;
pub extern fn MsneGroupContext(group: *Group, member: u32) *Ctx;
pub extern fn MsneGroupLoadGlb(group: *Group, glb_path: [*:0]const u8, info: ?*GlbInfo) c_int;
pub extern fn MsneGroupSetBackgroundExr(group: *Group, exr_path: [*:0]const u8) c_int;
pub extern fn MsneGroupSetPipeline(group: *Group, opts: *const PipelineOpts) c_int;
pub extern fn MsneGroupCreateSensor(group: *Group, extent: Extent2D) i64;
pub extern fn MsneGroupRender(group: *Group, sensor: u32, lens: u32, launches: u32) c_int;
pub extern fn MsneGroupRenderProgressive(group: *Group, sensor: u32, lens: u32, frames: u32, max_sample_count: u32, gather_every: u32, present: ?PresentFn, user: ?*anyopaque) c_int;
pub extern fn MsneGroupGetStats(group: *Group, summed: *Stats, gather_ms: ?*f64, gathers: ?*u64) c_int;
pub extern fn MsneGroupTransport(group: *const Group) [*:0]const u8;
pub extern fn MsneGroupGetLastError(group: ?*const Group) [*:0]const u8;

pub const Error = error{ NoDevice, CallFailed };

/// the error message of a failed call on stderr, then error.CallFailed
pub fn check(ctx: ?*const Ctx, rc: anytype) Error!void {
    const failed = switch (@typeInfo(@TypeOf(rc))) { .Bool => !rc, else => rc < 0 or (@TypeOf(rc) == c_int and rc != 0) };
    if (!failed) return;
    std.debug.print("moonshine_amd: {s}\n", .{std.mem.span(MsneGetLastError(ctx))});
    return Error.CallFailed;
}
