//! offline.zig — the reference's headless renderer (offline/main.zig:19-203) with its GPU half routed through libmoonshine_amd.so:
//!     offline <in.glb> <skybox.exr> <out.exr> [spp=16] [gpus=1]
//! Same positional arguments, extension checks and defaults as offline/main.zig:27-50 (1280x720, 16 spp), the same pipeline
//! constants (:106-111: samples_per_run 1, max_bounces 1024, one env + one mesh light sample per bounce) and the same timing lines
//! (:59-76) — as behaviour, in this file's own words.  What the reference records into one command buffer — `spp` x (push constants, trace rays, barrier), a
//! copy to a host buffer (:120-186) — is ONE call here: MsneRender(ctx, sensor, lens, spp, readback = 1); Rgba2D.save (:190)
//! is MsneSaveSensorExr.  With gpus > 1 the image tiles shard over that many GPUs in this process (MsneGroup: one context per
//! GPU, one RCCL gather of the films) — no reference equivalent, VulkanContext.zig:313-326 picks one device.
//!
//! NOT COMPILED in the image this was written in (no Zig toolchain there); host/offline.cpp is the same program in C++ and is what
//! tests/test_gpu_io.py::test_offline_cli runs.  Build:  zig build-exe offline.zig -lc -L../.. -lmoonshine_amd -rpath ../..
const std = @import("std");
const amd = @import("amd.zig");

// The CLI CONTRACT of offline/main.zig:27-50 — <in.glb> <skybox.exr> <out.exr> [spp = 16], the three extension checks, 1280x720 — and its timing lines (:59-76:
// "<s>.<ms> seconds to <state>") are kept as behaviour; the text is this file's own (an arena holds the arguments, a std.time.Timer the clock).  A maintainer who
// integrates into the reference keeps THEIR Config / IntervalLogger and takes only main's body below: INTEGRATION.md section 3 shows that as a diff.
const Args = struct {
    glb: [:0]const u8,
    skybox: [:0]const u8,
    out: [:0]const u8,
    spp: u32 = 16,
    gpus: u32 = 1,
    extent: amd.Extent2D = .{ .width = 1280, .height = 720 },

    fn wants(path: []const u8, ext: []const u8, err: anyerror) !void {
        if (!std.mem.endsWith(u8, path, ext) or !std.mem.eql(u8, std.fs.path.extension(path), ext)) return err;
    }

    fn parse(arena: std.mem.Allocator) !Args {
        var it = try std.process.argsWithAllocator(arena);
        _ = it.skip();
        var a = Args{
            .glb = try arena.dupeZ(u8, it.next() orelse return error.BadArgs),
            .skybox = try arena.dupeZ(u8, it.next() orelse return error.BadArgs),
            .out = try arena.dupeZ(u8, it.next() orelse return error.BadArgs),
        };
        try wants(a.glb, ".glb", error.OnlySupportsGlbInput);
        try wants(a.skybox, ".exr", error.OnlySupportsExrSkybox);
        try wants(a.out, ".exr", error.OnlySupportsExrOutput);
        if (it.next()) |n| a.spp = try std.fmt.parseInt(u32, n, 10);
        if (it.next()) |n| a.gpus = try std.fmt.parseInt(u32, n, 10);
        return a;
    }
};

const Laps = struct {
    timer: std.time.Timer,

    fn lap(self: *Laps, what: []const u8) !void {
        const ms = self.timer.lap() / std.time.ns_per_ms;
        try std.io.getStdOut().writer().print("{}.{:0>3} seconds to {s}\n", .{ ms / 1000, ms % 1000, what });
    }
};

const pipeline_opts = amd.PipelineOpts{ .samples_per_run = 1, .max_bounces = 1024, .env_samples_per_bounce = 1, .mesh_samples_per_bounce = 1 }; // offline/main.zig:106-111

fn ioFailed(what: []const u8) error{IoFailed} {
    std.debug.print("{s}: {s}\n", .{ what, std.mem.span(amd.MsneGetIoError()) });
    return error.IoFailed;
}

pub fn main() !void {
    var logger = Laps{ .timer = try std.time.Timer.start() };

    var arena = std.heap.ArenaAllocator.init(std.heap.page_allocator);
    defer arena.deinit();
    const config = try Args.parse(arena.allocator());

    if (config.gpus > 1) return renderOnGroup(config, &logger);

    const ctx = amd.MsneCreate(null) orelse { // VulkanContext.create + VkAllocator + Commands (:91-98)
        std.debug.print("moonshine_amd: {s}\n", .{std.mem.span(amd.MsneGetLastError(null))});
        return error.NoDevice;
    };
    defer amd.HdMoonshineDestroy(ctx);
    try logger.lap("set up initial state");

    // Scene.fromGlbExr (:102): world + first camera from the glb, background from the exr, one sensor of `extent`
    var info: amd.GlbInfo = undefined;
    if (amd.MsneLoadGlb(ctx, config.glb.ptr, &info) != 0) return ioFailed("glb");
    if (amd.MsneSetBackgroundExr(ctx, config.skybox.ptr) != 0) return ioFailed("skybox");
    const sensor = amd.HdMoonshineCreateSensor(ctx, config.extent);
    try logger.lap("load world");

    try amd.check(ctx, amd.MsneSetPipeline(ctx, &pipeline_opts)); // Pipeline.create (:106-112): constants are kernel arguments, nothing is compiled
    try logger.lap("create pipeline");

    // :120-186 — spp launches, each adding one sample per pixel to the running mean, then the film in host memory
    try amd.check(ctx, amd.MsneRender(ctx, sensor, info.lens, config.spp, 1));
    try logger.lap("render");

    if (amd.MsneSaveSensorExr(ctx, sensor, config.extent, config.out.ptr) != 0) return ioFailed("exr"); // Rgba2D.save (:190)
    try logger.lap("write exr");
}

fn renderOnGroup(config: Args, logger: *Laps) !void {
    const group = amd.MsneGroupCreate(null, config.gpus, 0) orelse {
        std.debug.print("moonshine_amd: {s}\n", .{std.mem.span(amd.MsneGroupGetLastError(null))});
        return error.NoDevice;
    };
    defer amd.MsneGroupDestroy(group);
    try logger.lap("set up initial state");

    var info: amd.GlbInfo = undefined;
    if (amd.MsneGroupLoadGlb(group, config.glb.ptr, &info) != 0) return ioFailed("glb"); // every member holds the whole scene
    if (amd.MsneGroupSetBackgroundExr(group, config.skybox.ptr) != 0) return ioFailed("skybox");
    const sensor_or_error = amd.MsneGroupCreateSensor(group, config.extent);
    if (sensor_or_error < 0) return error.CallFailed;
    const sensor: u32 = @intCast(sensor_or_error);
    try logger.lap("load world");

    if (amd.MsneGroupSetPipeline(group, &pipeline_opts) != 0) return error.CallFailed;
    try logger.lap("create pipeline");

    if (amd.MsneGroupRender(group, sensor, info.lens, config.spp) != 0) { // every member its tiles, one gather, unpack on member 0
        std.debug.print("moonshine_amd: {s}\n", .{std.mem.span(amd.MsneGroupGetLastError(group))});
        return error.CallFailed;
    }
    try logger.lap("render");

    const root = amd.MsneGroupContext(group, 0);
    if (amd.MsneSaveSensorExr(root, sensor, config.extent, config.out.ptr) != 0) return ioFailed("exr");
    try logger.lap("write exr");
}
