"""Procedural scenes for the hot path (host-side data generation only — no rendering here).

* icosphere(): the reference's only geometry fixture, engine/tests.zig:115-247, f32 arithmetic reproduced.
* furnace_*: the reference's two live furnace tests, engine/tests.zig:257-455.
* s1 / s2 / cornell: the synthetic workloads of SURVEY.md §8(d) / BASELINE.md §2.2.

Every builder takes a `ctx` exposing the Context surface (moonshine_amd.api.Context for the
HIP path; tests also pass the oracle's binding) and returns (sensor, lens) handles.
"""
import math

import numpy as np

GLASS, LAMBERT, PERFECT_MIRROR, STANDARD_PBR = 0, 1, 2, 3
F = np.float32


def icosphere(order, reverse_winding_order=False):
    """engine/tests.zig:115-247.  Midpoints are taken between *unnormalised* f32 positions and
    every position is normalised once at the end (tests.zig:226-230)."""
    t = F((1.0 + math.sqrt(5.0)) / 2.0)
    o, z = F(1), F(0)
    pos = [(-o, t, z), (o, t, z), (-o, -t, z), (o, -t, z), (z, -o, t), (z, o, t), (z, -o, -t), (z, o, -t),
           (t, z, -o), (t, z, o), (-t, z, -o), (-t, z, o)]
    pos = [tuple(F(c) for c in p) for p in pos]
    tris = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
            (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
            (8, 6, 7), (9, 8, 1)]
    cache = {}

    def midpoint(i1, i2):
        key = (min(i1, i2), max(i1, i2))
        if key in cache:
            return cache[key]
        p1, p2 = pos[i1], pos[i2]
        pos.append(tuple(F(F(a + b) / F(2.0)) for a, b in zip(p1, p2)))
        cache[key] = len(pos) - 1
        return cache[key]

    for _ in range(order):
        nxt = []
        for (x, y, zz) in tris:
            a = midpoint(x, y); b = midpoint(y, zz); c = midpoint(zz, x)
            nxt += [(x, a, c), (y, b, a), (zz, c, b), (a, b, c)]
        tris = nxt
    P = np.array(pos, dtype=np.float32)
    # unit(): div_scalar(length) with length = sqrt(x*x + y*y + z*z) in f32 (vector.zig:123-148)
    l = np.sqrt((P[:, 0] * P[:, 0] + P[:, 1] * P[:, 1]) + P[:, 2] * P[:, 2], dtype=np.float32)
    P = (P / l[:, None]).astype(np.float32)
    I = np.array(tris, dtype=np.uint32)
    if reverse_winding_order:
        I = I[:, ::-1].copy()
    return P, I


def _lens(origin, forward, up, vfov, aperture=0.0, focus=1.0):
    return dict(origin=origin, forward=forward, up=up, vfov=vfov, aperture=aperture, focus_distance=focus)


def furnace_white_sphere(ctx, extent=(32, 32), order=5):
    """tests.zig:257-344: white Lambert icosphere(5) in a 1x1 white environment."""
    P, I = icosphere(order, False)
    mesh = ctx.create_mesh(P, I)
    normal = ctx.solid_texture(0.5, 0.5)
    albedo = ctx.solid_texture(1.0, 1.0, 1.0)
    emissive = ctx.solid_texture(0.0, 0.0, 0.0)
    mat = ctx.create_material(LAMBERT, normal, emissive, color=albedo)
    ctx.create_instance([(mesh, mat, False)])
    lens = ctx.create_lens(ctx.make_lens(**_lens((-3, 0, 0), (1, 0, 0), (0, 0, 1), math.pi / 4.0)))
    sensor = ctx.create_sensor(*extent)
    ctx.set_background(np.array([1, 1, 1, 1], np.float32), 1, 1)
    return sensor, lens


def furnace_sphere(ctx, kind, extent=(32, 32), order=5, color=(1.0, 1.0, 1.0), metalness=0.0, roughness=0.5, ior=1.5, env="white"):
    """The reference's furnace shape (tests.zig:257-344: unit icosphere(5), lens at (-3,0,0), 1x1 white environment) with the
    sphere's material swapped: the reference only runs Lambert through it; mirror and glass have f*|cos|/pdf == 1, so every
    pixel must still come out as the environment's 1.0, and StandardPBR must not exceed it (energy conservation)."""
    P, I = icosphere(order, False)
    mesh = ctx.create_mesh(P, I)
    normal = ctx.solid_texture(0.5, 0.5)
    black = ctx.solid_texture(0.0, 0.0, 0.0)
    if kind == STANDARD_PBR:
        mat = ctx.create_material(STANDARD_PBR, normal, black, color=ctx.solid_texture(*color), metalness=ctx.solid_texture(metalness),
                                  roughness=ctx.solid_texture(roughness), ior=ior)
    elif kind == LAMBERT:
        mat = ctx.create_material(LAMBERT, normal, black, color=ctx.solid_texture(*color))
    else:
        mat = ctx.create_material(kind, normal, black, ior=ior)
    ctx.create_instance([(mesh, mat, False)])
    lens = ctx.create_lens(ctx.make_lens(**_lens((-3, 0, 0), (1, 0, 0), (0, 0, 1), math.pi / 4.0)))
    sensor = ctx.create_sensor(*extent)
    if env == "white":
        ctx.set_background(np.array([1, 1, 1, 1], np.float32), 1, 1)
    else:
        img = sky_sun_equirect()
        ctx.set_background(img, img.shape[1], img.shape[0])
    return sensor, lens


def furnace_inside_sphere(ctx, extent=(32, 32), order=5, sampled=False):
    """tests.zig:366-455: camera inside a reversed-winding icosphere, albedo 0.5, emissive 0.5, black env.
    sampled=True: the sphere is a mesh light (the reference's disabled variant, tests.zig:457-487)."""
    P, I = icosphere(order, True)
    mesh = ctx.create_mesh(P, I)
    normal = ctx.solid_texture(0.5, 0.5)
    albedo = ctx.solid_texture(0.5, 0.5, 0.5)
    emissive = ctx.solid_texture(0.5, 0.5, 0.5)
    mat = ctx.create_material(LAMBERT, normal, emissive, color=albedo)
    ctx.create_instance([(mesh, mat, sampled)])
    lens = ctx.create_lens(ctx.make_lens(**_lens((0, 0, 0), (1, 0, 0), (0, 0, 1), math.pi / 3.0)))
    sensor = ctx.create_sensor(*extent)
    ctx.set_background(np.array([0, 0, 0, 1], np.float32), 1, 1)
    return sensor, lens


def quad(p0, p1, p2, p3):
    P = np.array([p0, p1, p2, p3], np.float32)
    I = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    return P, I


def sky_sun_equirect(w=512, h=256):
    """L = (0.2,0.3,0.5)*max(cos(theta),0) + 50*exp(-(angle to (1,1,1)/sqrt3)^2 / 0.002)."""
    theta = (np.arange(h, dtype=np.float64) + 0.5) / h * math.pi
    phi = (np.arange(w, dtype=np.float64) + 0.5) / w * 2.0 * math.pi
    T, Ph = np.meshgrid(theta, phi, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(Ph), np.sin(T) * np.sin(Ph), np.cos(T)], -1)
    sun = np.array([1.0, 1.0, 1.0]) / math.sqrt(3.0)
    ang = np.arccos(np.clip(d @ sun, -1.0, 1.0))
    base = np.maximum(np.cos(T), 0.0)[..., None] * np.array([0.2, 0.3, 0.5])
    img = base + (50.0 * np.exp(-(ang ** 2) / 0.002))[..., None]
    out = np.ones((h, w, 4), np.float32)
    out[..., :3] = img.astype(np.float32)
    return out


def s1(ctx, extent=(1920, 1080), grid=7, order=5, env="constant"):
    """The north-star workload, SURVEY.md §8(d) #S1: grid x grid order-`order` icospheres (7x7x20480 = 1 003 520
    triangles) as distinct baked meshes + ground quad + 4x4 emissive quad "Emitter" at z=8."""
    P0, I0 = icosphere(order, False)
    normal = ctx.solid_texture(0.5, 0.5)
    black = ctx.solid_texture(0.0, 0.0, 0.0)
    mats = [
        ctx.create_material(LAMBERT, normal, black, color=ctx.solid_texture(0.7, 0.7, 0.7)),
        ctx.create_material(STANDARD_PBR, normal, black, color=ctx.solid_texture(0.9, 0.6, 0.2),
                            metalness=ctx.solid_texture(1.0), roughness=ctx.solid_texture(0.3), ior=1.5),
        ctx.create_material(GLASS, normal, black, ior=1.5),
        ctx.create_material(PERFECT_MIRROR, normal, black),
    ]
    half = (grid - 1) / 2.0
    k = 0
    for gy in range(grid):
        for gx in range(grid):
            off = np.array([(gx - half) * 2.5, (gy - half) * 2.5, 1.0], np.float32)
            mesh = ctx.create_mesh((P0 + off).astype(np.float32), I0)
            ctx.create_instance([(mesh, mats[k % 4], False)])
            k += 1
    e = half * 2.5 + 6.0
    gP, gI = quad((-e, -e, 0), (e, -e, 0), (e, e, 0), (-e, e, 0))
    ground = ctx.create_material(LAMBERT, normal, black, color=ctx.solid_texture(0.8, 0.8, 0.8))
    ctx.create_instance([(ctx.create_mesh(gP, gI), ground, False)])
    # emitter faces -z: winding chosen so cross(p0-p2, p1-p2) points down
    lP, lI = quad((-2, -2, 8), (-2, 2, 8), (2, 2, 8), (2, -2, 8))
    emitter = ctx.create_material(LAMBERT, normal, ctx.solid_texture(10.0, 10.0, 10.0), color=black)
    ctx.create_instance([(ctx.create_mesh(lP, lI), emitter, True)])
    if env == "constant":
        ctx.set_background(np.array([0.5, 0.5, 0.5, 1], np.float32), 1, 1)
    else:
        img = sky_sun_equirect()
        ctx.set_background(img, img.shape[1], img.shape[0])
    f = np.array([14, 14, -8], np.float64); f /= np.linalg.norm(f)
    lens = ctx.create_lens(ctx.make_lens(**_lens((-14, -14, 9), tuple(f.astype(np.float32)), (0, 0, 1), 0.6)))
    sensor = ctx.create_sensor(*extent)
    return sensor, lens


def _rot(axis, angle):
    a = np.asarray(axis, np.float64); a /= np.linalg.norm(a)
    c, s = math.cos(angle), math.sin(angle)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) * c + s * K + (1 - c) * np.outer(a, a)


def s2(ctx, extent=(1920, 1080), dims=(10, 10, 5), order=5):
    """SURVEY.md §8(d) #S2: ONE icosphere mesh, dims.x*dims.y*dims.z instances with LCG scale/rotation,
    alternating Glass(1.5) / StandardPBR(metalness .5, roughness .2)."""
    P0, I0 = icosphere(order, False)
    mesh = ctx.create_mesh(P0, I0)
    normal = ctx.solid_texture(0.5, 0.5)
    black = ctx.solid_texture(0.0, 0.0, 0.0)
    mats = [ctx.create_material(GLASS, normal, black, ior=1.5),
            ctx.create_material(STANDARD_PBR, normal, black, color=ctx.solid_texture(0.8, 0.8, 0.8),
                                metalness=ctx.solid_texture(0.5), roughness=ctx.solid_texture(0.2), ior=1.5)]
    state = 12345
    def lcg():
        nonlocal state
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        return state / 4294967296.0
    k = 0
    for iz in range(dims[2]):
        for iy in range(dims[1]):
            for ix in range(dims[0]):
                sc = 0.6 + 0.4 * lcg()
                R = _rot((lcg() - 0.5, lcg() - 0.5, lcg() - 0.5 + 1e-3), lcg() * 2 * math.pi) * sc
                T = np.zeros((3, 4), np.float32)
                T[:, :3] = R.astype(np.float32)
                T[:, 3] = [(ix - (dims[0] - 1) / 2) * 2.5, (iy - (dims[1] - 1) / 2) * 2.5, 1.0 + iz * 2.5]
                ctx.create_instance([(mesh, mats[k % 2], False)], transform=T)
                k += 1
    e = max(dims[0], dims[1]) * 1.25 + 6.0
    gP, gI = quad((-e, -e, 0), (e, -e, 0), (e, e, 0), (-e, e, 0))
    ground = ctx.create_material(LAMBERT, normal, black, color=ctx.solid_texture(0.8, 0.8, 0.8))
    ctx.create_instance([(ctx.create_mesh(gP, gI), ground, False)])
    top = 1.0 + dims[2] * 2.5 + 4.0
    lP, lI = quad((-3, -3, top), (-3, 3, top), (3, 3, top), (3, -3, top))
    emitter = ctx.create_material(LAMBERT, normal, ctx.solid_texture(10.0, 10.0, 10.0), color=black)
    ctx.create_instance([(ctx.create_mesh(lP, lI), emitter, True)])
    ctx.set_background(np.array([0.5, 0.5, 0.5, 1], np.float32), 1, 1)
    o = np.array([-e - 8, -e - 8, top * 0.8]); f = -o + np.array([0, 0, top * 0.3]); f /= np.linalg.norm(f)
    lens = ctx.create_lens(ctx.make_lens(**_lens(tuple(o.astype(np.float32)), tuple(f.astype(np.float32)), (0, 0, 1), 0.6)))
    sensor = ctx.create_sensor(*extent)
    return sensor, lens


def cornell(ctx, extent=(512, 512)):
    """BASELINE.json configs[1]: Cornell box, emissive ceiling quad ("Emitter", sampled), Lambert walls + 2 boxes."""
    normal = ctx.solid_texture(0.5, 0.5)
    black = ctx.solid_texture(0.0, 0.0, 0.0)
    def lam(r, g, b):
        return ctx.create_material(LAMBERT, normal, black, color=ctx.solid_texture(r, g, b))
    white, red, green = lam(0.73, 0.73, 0.73), lam(0.65, 0.05, 0.05), lam(0.12, 0.45, 0.15)
    def add(P, I, m, sampled=False):
        ctx.create_instance([(ctx.create_mesh(P, I), m, sampled)])
    # room [-1,1]^2 x [0,2], open towards -y (camera side); normals point inwards
    add(*quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)), white)          # floor (+z)
    add(*quad((-1, -1, 2), (-1, 1, 2), (1, 1, 2), (1, -1, 2)), white)          # ceiling (-z)
    add(*quad((-1, 1, 0), (1, 1, 0), (1, 1, 2), (-1, 1, 2)), white)            # back (-y)
    add(*quad((-1, -1, 0), (-1, 1, 0), (-1, 1, 2), (-1, -1, 2)), red)          # left (+x)
    add(*quad((1, -1, 0), (1, -1, 2), (1, 1, 2), (1, 1, 0)), green)            # right (-x)
    def box(cx, cy, sx, sy, h, ang):
        c, s = math.cos(ang), math.sin(ang)
        def p(x, y, z):
            return (cx + c * x - s * y, cy + s * x + c * y, z)
        v = [p(-sx, -sy, 0), p(sx, -sy, 0), p(sx, sy, 0), p(-sx, sy, 0), p(-sx, -sy, h), p(sx, -sy, h), p(sx, sy, h), p(-sx, sy, h)]
        P = np.array(v, np.float32)
        I = np.array([[4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5], [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], np.uint32)
        return P, I
    add(*box(-0.35, 0.3, 0.3, 0.3, 1.2, 0.3), white)
    add(*box(0.35, -0.3, 0.3, 0.3, 0.6, -0.3), white)
    emitter = ctx.create_material(LAMBERT, normal, ctx.solid_texture(17.0, 12.0, 4.0), color=black)
    add(*quad((-0.25, -0.25, 1.99), (-0.25, 0.25, 1.99), (0.25, 0.25, 1.99), (0.25, -0.25, 1.99)), emitter, True)
    ctx.set_background(np.array([0, 0, 0, 1], np.float32), 1, 1)
    lens = ctx.create_lens(ctx.make_lens(**_lens((0, -3.9, 1), (0, 1, 0), (0, 0, 1), 0.69)))
    sensor = ctx.create_sensor(*extent)
    return sensor, lens


def single_triangle(ctx, extent=(64, 64)):
    """BASELINE.json configs[0]: one Lambert triangle + constant environment."""
    P = np.array([(-1, 0, -1), (1, 0, -1), (0, 0, 1)], np.float32)
    I = np.array([[0, 1, 2]], np.uint32)
    normal = ctx.solid_texture(0.5, 0.5)
    black = ctx.solid_texture(0.0, 0.0, 0.0)
    m = ctx.create_material(LAMBERT, normal, black, color=ctx.solid_texture(0.8, 0.3, 0.3))
    ctx.create_instance([(ctx.create_mesh(P, I), m, False)])
    ctx.set_background(np.array([1, 1, 1, 1], np.float32), 1, 1)
    lens = ctx.create_lens(ctx.make_lens(**_lens((0, -3, 0), (0, 1, 0), (0, 0, 1), 0.8)))
    sensor = ctx.create_sensor(*extent)
    return sensor, lens
