//! furnace_test.zig — the reference's furnace tests (engine/tests.zig:257-487) against libmoonshine_amd.so through amd.zig, at the
//! reference's own parameters and tolerances.  tests/test_gpu_parity.py::test_furnace_* are these tests driven through ctypes (and
//! additionally compare every pixel with the oracle, bit for bit); this file is what a maintainer of the reference would run:
//!     zig test furnace_test.zig -lc -L../.. -lmoonshine_amd -rpath ../..
//! NOT COMPILED in the image this was written in (no Zig toolchain there).
const std = @import("std");
const amd = @import("amd.zig");
const F32x3 = amd.F32x3;
const U32x3 = amd.U32x3;

const Mesh = struct {
    positions: []F32x3,
    indices: []U32x3,
    fn destroy(self: Mesh, allocator: std.mem.Allocator) void {
        allocator.free(self.positions);
        allocator.free(self.indices);
    }
};

// engine/tests.zig:110-246 — a unit icosphere at the origin: the icosahedron of :179-215 subdivided `order` times with a midpoint
// cache (vertex order as the reference creates it: new midpoints are appended), projected onto the sphere afterwards
fn icosphere(order: usize, allocator: std.mem.Allocator, reverse_winding_order: bool) !Mesh {
    const t: f32 = (1.0 + @sqrt(5.0)) / 2.0;
    var positions = std.ArrayList(F32x3).init(allocator);
    defer positions.deinit();
    var triangles = std.ArrayList(U32x3).init(allocator);
    defer triangles.deinit();
    try positions.appendSlice(&[12]F32x3{
        F32x3.new(-1, t, 0), F32x3.new(1, t, 0),  F32x3.new(-1, -t, 0), F32x3.new(1, -t, 0),
        F32x3.new(0, -1, t), F32x3.new(0, 1, t),  F32x3.new(0, -1, -t), F32x3.new(0, 1, -t),
        F32x3.new(t, 0, -1), F32x3.new(t, 0, 1),  F32x3.new(-t, 0, -1), F32x3.new(-t, 0, 1),
    });
    try triangles.appendSlice(&[20]U32x3{
        U32x3.new(0, 11, 5), U32x3.new(0, 5, 1),  U32x3.new(0, 1, 7),   U32x3.new(0, 7, 10), U32x3.new(0, 10, 11),
        U32x3.new(1, 5, 9),  U32x3.new(5, 11, 4), U32x3.new(11, 10, 2), U32x3.new(10, 7, 6), U32x3.new(7, 1, 8),
        U32x3.new(3, 9, 4),  U32x3.new(3, 4, 2),  U32x3.new(3, 2, 6),   U32x3.new(3, 6, 8),  U32x3.new(3, 8, 9),
        U32x3.new(4, 9, 5),  U32x3.new(2, 4, 11), U32x3.new(6, 2, 10),  U32x3.new(8, 6, 7),  U32x3.new(9, 8, 1),
    });
    var cache = std.AutoArrayHashMap(u64, u32).init(allocator);
    defer cache.deinit();
    for (0..order) |_| {
        var next = std.ArrayList(U32x3).init(allocator);
        errdefer next.deinit();
        for (triangles.items) |tri| {
            const corners = [3][2]u32{ .{ tri.x, tri.y }, .{ tri.y, tri.z }, .{ tri.z, tri.x } };
            var mid: [3]u32 = undefined;
            for (corners, 0..) |e, k| {
                const smaller = @min(e[0], e[1]);
                const greater = @max(e[0], e[1]);
                const key = (@as(u64, smaller) << 32) + greater;
                if (cache.get(key)) |found| {
                    mid[k] = found;
                } else {
                    try positions.append(positions.items[e[0]].add(positions.items[e[1]]).div_scalar(2.0));
                    mid[k] = @intCast(positions.items.len - 1);
                    try cache.put(key, mid[k]);
                }
            }
            try next.append(U32x3.new(tri.x, mid[0], mid[2]));
            try next.append(U32x3.new(tri.y, mid[1], mid[0]));
            try next.append(U32x3.new(tri.z, mid[2], mid[1]));
            try next.append(U32x3.new(mid[0], mid[1], mid[2]));
        }
        triangles.deinit();
        triangles = next;
    }
    const out_positions = try allocator.dupe(F32x3, positions.items);
    for (out_positions) |*p| p.* = p.unit();
    const out_indices = try allocator.dupe(U32x3, triangles.items);
    if (reverse_winding_order) for (out_indices) |*i| { i.* = U32x3.new(i.z, i.y, i.x); };
    return Mesh{ .positions = out_positions, .indices = out_indices };
}

const default_normal = amd.F32x2.new(0.5, 0.5); // MaterialManager.MaterialInfo.default_normal

/// the scene every furnace test builds: one order-5 icosphere with a Lambert material, one lens, a 32 x 32 sensor, a constant background
fn renderSphere(allocator: std.mem.Allocator, reverse: bool, albedo: f32, emissive: f32, sampled: bool, background: [4]f32, lens: amd.Lens, opts: amd.PipelineOpts) ![]const [4]f32 {
    const ctx = amd.MsneCreate(null) orelse return error.NoDevice; // TestingContext.create (tests.zig:41-66)
    errdefer amd.HdMoonshineDestroy(ctx);
    const sphere = try icosphere(5, allocator, reverse);
    defer sphere.destroy(allocator);
    const mesh: u32 = @intCast(amd.MsneCreateMesh(ctx, sphere.positions.ptr, null, null, sphere.positions.len, 0, sphere.indices.ptr, sphere.indices.len));
    const normal_texture = amd.HdMoonshineCreateSolidTexture2(ctx, default_normal, "");
    const albedo_texture = amd.HdMoonshineCreateSolidTexture3(ctx, F32x3.new(albedo, albedo, albedo), "");
    const emissive_texture = amd.HdMoonshineCreateSolidTexture3(ctx, F32x3.new(emissive, emissive, emissive), "");
    const material: u32 = @intCast(amd.MsneCreateMaterial(ctx, &.{ .normal = normal_texture, .emissive = emissive_texture, .type = .lambert, .color = albedo_texture }));
    _ = amd.HdMoonshineCreateInstance(ctx, amd.Mat3x4.identity, &[1]amd.Geometry{.{ .mesh = mesh, .material = material, .sampled = sampled }}, 1, true);
    const lens_handle = amd.HdMoonshineCreateLens(ctx, lens);
    const sensor = amd.HdMoonshineCreateSensor(ctx, .{ .width = 32, .height = 32 });
    var bg = background;
    try amd.check(ctx, amd.MsneSetBackground(ctx, &bg, .{ .width = 1, .height = 1 }));
    try amd.check(ctx, amd.MsneSetPipeline(ctx, &opts));
    try amd.check(ctx, amd.MsneRender(ctx, sensor, lens_handle, 1, 1)); // tc.renderToOutput (tests.zig:68-101): one launch of samples_per_run samples
    const out = try allocator.dupe([4]f32, amd.HdMoonshineGetSensorData(ctx, sensor)[0 .. 32 * 32]);
    amd.HdMoonshineDestroy(ctx);
    return out;
}

fn expectWhite(pixels: []const [4]f32, tolerance: f32) !void {
    for (pixels) |pixel| for (pixel[0..3]) |component| {
        if (!std.math.approxEqAbs(f32, component, 1.0, tolerance)) return error.NonWhitePixel;
    };
}

const outside = amd.Lens{ .origin = F32x3.new(-3, 0, 0), .forward = F32x3.new(1, 0, 0), .up = F32x3.new(0, 0, 1), .vfov = std.math.pi / 4.0, .aperture = 0, .focus_distance = 1 };
const inside = amd.Lens{ .origin = F32x3.new(0, 0, 0), .forward = F32x3.new(1, 0, 0), .up = F32x3.new(0, 0, 1), .vfov = std.math.pi / 3.0, .aperture = 0, .focus_distance = 1 };

test "white sphere on white background is white" { // tests.zig:257-344
    const a = std.testing.allocator;
    const pixels = try renderSphere(a, false, 1.0, 0.0, false, .{ 1, 1, 1, 1 }, outside, .{ .samples_per_run = 512, .max_bounces = 1024, .env_samples_per_bounce = 0, .mesh_samples_per_bounce = 0 });
    defer a.free(pixels);
    try expectWhite(pixels, 0.00001);
}

test "white sphere on white background is white with env sampling" { // tests.zig:346-364 (same scene, env_samples_per_bounce = 1)
    const a = std.testing.allocator;
    const pixels = try renderSphere(a, false, 1.0, 0.0, false, .{ 1, 1, 1, 1 }, outside, .{ .samples_per_run = 512, .max_bounces = 1024, .env_samples_per_bounce = 1, .mesh_samples_per_bounce = 0 });
    defer a.free(pixels);
    try expectWhite(pixels, 0.1);
}

test "inside illuminating sphere is white" { // tests.zig:366-455: albedo 0.5 + emission 0.5 -> radiance 1 from inside
    const a = std.testing.allocator;
    const pixels = try renderSphere(a, true, 0.5, 0.5, false, .{ 0, 0, 0, 1 }, inside, .{ .samples_per_run = 1024, .max_bounces = 1024, .env_samples_per_bounce = 0, .mesh_samples_per_bounce = 0 });
    defer a.free(pixels);
    try expectWhite(pixels, 0.02);
}

test "inside illuminating sphere is white with mesh sampling" { // tests.zig:457-487 (commented out in the reference; runs here)
    const a = std.testing.allocator;
    const pixels = try renderSphere(a, true, 0.5, 0.5, true, .{ 0, 0, 0, 1 }, inside, .{ .samples_per_run = 512, .max_bounces = 1024, .env_samples_per_bounce = 0, .mesh_samples_per_bounce = 1 });
    defer a.free(pixels);
    try expectWhite(pixels, 0.1);
}
