//! offline.zig — the reference's headless renderer (offline/main.zig:19-203) with its GPU half routed through libmoonshine_amd.so:
//!     offline <in.glb> <skybox.exr> <out.exr> [spp=16] [gpus=1]
//! Same positional arguments, extension checks and defaults as offline/main.zig:27-50 (1280x720, 16 spp), the same pipeline
//! constants (:106-111: samples_per_run 1, max_bounces 1024, one env + one mesh light sample per bounce) and the same timing lines
//! (IntervalLogger, :59-76).  What the reference records into one command buffer — `spp` x (push constants, trace rays, barrier), a
//! copy to a host buffer (:120-186) — is ONE call here: MsneRender(ctx, sensor, lens, spp, readback = 1); Rgba2D.save (:190)
//! is MsneSaveSensorExr.  With gpus > 1 the image tiles shard over that many GPUs in this process (MsneGroup: one context per
//! GPU, one RCCL gather of the films) — no reference equivalent, VulkanContext.zig:313-326 picks one device.
//!
//! NOT COMPILED in the image this was written in (no Zig toolchain there); host/offline.cpp is the same program in C++ and is what
//! tests/test_gpu_io.py::test_offline_cli runs.  Build:  zig build-exe offline.zig -lc -L../.. -lmoonshine_amd -rpath ../..
const std = @import("std");
const amd = @import("amd.zig");

const Config = struct {
    in_filepath: [:0]const u8, // must be glb
    out_filepath: [:0]const u8, // must be exr
    skybox_filepath: [:0]const u8, // must be exr
    spp: u32,
    gpus: u32,
    extent: amd.Extent2D,

    fn fromCli(allocator: std.mem.Allocator) !Config {
        const args = try std.process.argsAlloc(allocator);
        defer std.process.argsFree(allocator, args);
        if (args.len < 4) return error.BadArgs;

        const in_filepath = args[1];
        if (!std.mem.eql(u8, std.fs.path.extension(in_filepath), ".glb")) return error.OnlySupportsGlbInput;

        const skybox_filepath = args[2];
        if (!std.mem.eql(u8, std.fs.path.extension(skybox_filepath), ".exr")) return error.OnlySupportsExrSkybox;

        const out_filepath = args[3];
        if (!std.mem.eql(u8, std.fs.path.extension(out_filepath), ".exr")) return error.OnlySupportsExrOutput;

        const spp = if (args.len > 4) try std.fmt.parseInt(u32, args[4], 10) else 16;
        const gpus = if (args.len > 5) try std.fmt.parseInt(u32, args[5], 10) else 1;

        return Config{
            .in_filepath = try allocator.dupeZ(u8, in_filepath),
            .out_filepath = try allocator.dupeZ(u8, out_filepath),
            .skybox_filepath = try allocator.dupeZ(u8, skybox_filepath),
            .spp = spp,
            .gpus = gpus,
            .extent = amd.Extent2D{ .width = 1280, .height = 720 }, // (offline/main.zig:47: "TODO: cli")
        };
    }

    fn destroy(self: Config, allocator: std.mem.Allocator) void {
        allocator.free(self.in_filepath);
        allocator.free(self.out_filepath);
        allocator.free(self.skybox_filepath);
    }
};

const IntervalLogger = struct { // offline/main.zig:59-76
    last_time: std.time.Instant,

    fn start() !IntervalLogger {
        return IntervalLogger{ .last_time = try std.time.Instant.now() };
    }

    fn log(self: *IntervalLogger, state: []const u8) !void {
        const new_time = try std.time.Instant.now();
        const elapsed = new_time.since(self.last_time);
        const ms = elapsed / std.time.ns_per_ms;
        const s = ms / std.time.ms_per_s;
        try std.io.getStdOut().writer().print("{}.{:0>3} seconds to {s}\n", .{ s, ms % std.time.ms_per_s, state });
        self.last_time = new_time;
    }
};

const pipeline_opts = amd.PipelineOpts{ .samples_per_run = 1, .max_bounces = 1024, .env_samples_per_bounce = 1, .mesh_samples_per_bounce = 1 }; // offline/main.zig:106-111

fn ioFailed(what: []const u8) error{IoFailed} {
    std.debug.print("{s}: {s}\n", .{ what, std.mem.span(amd.MsneGetIoError()) });
    return error.IoFailed;
}

pub fn main() !void {
    var logger = try IntervalLogger.start();

    var gpa = std.heap.GeneralPurposeAllocator(.{}){};
    defer _ = gpa.deinit();
    const allocator = gpa.allocator();

    const config = try Config.fromCli(allocator);
    defer config.destroy(allocator);

    if (config.gpus > 1) return renderOnGroup(config, &logger);

    const ctx = amd.MsneCreate(null) orelse { // VulkanContext.create + VkAllocator + Commands (:91-98)
        std.debug.print("moonshine_amd: {s}\n", .{std.mem.span(amd.MsneGetLastError(null))});
        return error.NoDevice;
    };
    defer amd.HdMoonshineDestroy(ctx);
    try logger.log("set up initial state");

    // Scene.fromGlbExr (:102): world + first camera from the glb, background from the exr, one sensor of `extent`
    var info: amd.GlbInfo = undefined;
    if (amd.MsneLoadGlb(ctx, config.in_filepath.ptr, &info) != 0) return ioFailed("glb");
    if (amd.MsneSetBackgroundExr(ctx, config.skybox_filepath.ptr) != 0) return ioFailed("skybox");
    const sensor = amd.HdMoonshineCreateSensor(ctx, config.extent);
    try logger.log("load world");

    try amd.check(ctx, amd.MsneSetPipeline(ctx, &pipeline_opts)); // Pipeline.create (:106-112): constants are kernel arguments, nothing is compiled
    try logger.log("create pipeline");

    // :120-186 — spp launches, each adding one sample per pixel to the running mean, then the film in host memory
    try amd.check(ctx, amd.MsneRender(ctx, sensor, info.lens, config.spp, 1));
    try logger.log("render");

    if (amd.MsneSaveSensorExr(ctx, sensor, config.extent, config.out_filepath.ptr) != 0) return ioFailed("exr"); // Rgba2D.save (:190)
    try logger.log("write exr");
}

fn renderOnGroup(config: Config, logger: *IntervalLogger) !void {
    const group = amd.MsneGroupCreate(null, config.gpus, 0) orelse {
        std.debug.print("moonshine_amd: {s}\n", .{std.mem.span(amd.MsneGroupGetLastError(null))});
        return error.NoDevice;
    };
    defer amd.MsneGroupDestroy(group);
    try logger.log("set up initial state");

    var info: amd.GlbInfo = undefined;
    if (amd.MsneGroupLoadGlb(group, config.in_filepath.ptr, &info) != 0) return ioFailed("glb"); // every member holds the whole scene
    if (amd.MsneGroupSetBackgroundExr(group, config.skybox_filepath.ptr) != 0) return ioFailed("skybox");
    const sensor_or_error = amd.MsneGroupCreateSensor(group, config.extent);
    if (sensor_or_error < 0) return error.CallFailed;
    const sensor: u32 = @intCast(sensor_or_error);
    try logger.log("load world");

    if (amd.MsneGroupSetPipeline(group, &pipeline_opts) != 0) return error.CallFailed;
    try logger.log("create pipeline");

    if (amd.MsneGroupRender(group, sensor, info.lens, config.spp) != 0) { // every member its tiles, one gather, unpack on member 0
        std.debug.print("moonshine_amd: {s}\n", .{std.mem.span(amd.MsneGroupGetLastError(group))});
        return error.CallFailed;
    }
    try logger.log("render");

    const root = amd.MsneGroupContext(group, 0);
    if (amd.MsneSaveSensorExr(root, sensor, config.extent, config.out_filepath.ptr) != 0) return ioFailed("exr");
    try logger.log("write exr");
}
