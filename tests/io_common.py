"""Shared helpers for the scene-I/O tests: asset generation (tests/assets.py) and the importer→oracle shim."""
import ctypes as C
import math
import os
import subprocess

import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))      # (tools/ import this module as tests.io_common)
import assets  # noqa: E402
from moonshine_amd import scenes  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SHIM = None


def shim(orc):
    """builds tests/shim/glb_oracle_shim.cpp + the product's host importers against liborc.so (test infrastructure)"""
    global _SHIM
    if _SHIM is None:
        orc.build()
        out = os.path.join(ROOT, "tests", "shim", "libglb_oracle_shim.so")
        srcs = [os.path.join(ROOT, "tests", "shim", "glb_oracle_shim.cpp")] + [os.path.join(ROOT, "moonshine_amd", "host", f) for f in ("glb.cpp", "png.cpp", "exr.cpp")]
        if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-comment", "-o", out] + srcs
                                  + ["-L" + os.path.join(ROOT, "oracle"), "-lorc", "-lz", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
        _SHIM = C.CDLL(out)
        _SHIM.ShimError.restype = C.c_char_p
        _SHIM.ShimLoadGlb.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        _SHIM.ShimSetBackgroundExr.argtypes = [C.c_void_p, C.c_char_p]
        _SHIM.ShimPngDecode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p]
    return _SHIM


def png_decode(orc, data):
    """the product's PNG decoder (moonshine_amd/host/png.cpp) -> (h, w, 3) uint8"""
    s = shim(orc)
    wh = (C.c_uint32 * 2)()
    assert s.ShimPngDecode(data, len(data), None, wh) == 0, s.ShimError()
    out = np.zeros((wh[1], wh[0], 3), np.uint8)
    assert s.ShimPngDecode(data, len(data), out.ctypes.data_as(C.c_void_p), wh) == 0, s.ShimError()
    return out


def oracle_load(orc, ctx, glb_path, exr_path=None):
    s = shim(orc)
    info = (C.c_uint32 * 6)()
    assert s.ShimLoadGlb(C.c_void_p(ctx.h), glb_path.encode(), info) == 0, s.ShimError()
    if exr_path:
        assert s.ShimSetBackgroundExr(C.c_void_p(ctx.h), exr_path.encode()) == 0, s.ShimError()
    return int(info[5]), dict(zip(("meshes", "materials", "instances", "textures", "triangles", "lens"), [int(x) for x in info]))


def write_single_triangle(path_glb, path_exr):
    """BASELINE.json configs[0]: single-triangle glTF + constant env."""
    b = assets.GlbBuilder()
    m = b.material("Red", base_color=(0.8, 0.3, 0.3), metallic=0.0, roughness=1.0)
    # Y-up object coordinates: the triangle stands in the x/y plane
    mesh = b.mesh([dict(positions=[(-1, -1, 0), (1, -1, 0), (0, 1, 0)], indices=[0, 1, 2], material=m)])
    b.node(mesh=mesh)
    b.node(camera=b.camera(0.8), matrix=assets.look_at_yup((0, 0, 3), (0, 0, 0)))
    open(path_glb, "wb").write(b.tobytes())
    open(path_exr, "wb").write(assets.exr_bytes(np.ones((1, 1, 4), np.float32)))


def write_gallery(path_glb, path_exr, u32=False, interleaved=False):
    """every import rule of World.zig in one file: Lambert / mirror / glass / constant PBR / textured PBR with normal +
    metallic-roughness + emissive maps, an "Emitter…" quad (sampled), node hierarchy with TRS + matrix, uv + normals."""
    rs = np.random.default_rng(11)
    b = assets.GlbBuilder(interleaved=interleaved)
    white = b.material("Floor", base_color=(0.7, 0.7, 0.7), metallic=0.0, roughness=1.0)
    mirror = b.material("Mirror", metallic=1.0, roughness=0.0)
    glass = b.material("Glass", transmission=1.0, ior=1.45)
    gold = b.material("Gold", base_color=(0.9, 0.6, 0.2), metallic=1.0, roughness=0.3)
    col = b.texture_png(rs.integers(0, 256, (16, 16, 3), dtype=np.uint8))
    mr = b.texture_png(rs.integers(20, 230, (8, 8, 3), dtype=np.uint8))
    nrm = b.texture_png(np.concatenate([128 + rs.integers(-30, 30, (8, 8, 2)), np.full((8, 8, 1), 255)], -1).astype(np.uint8))
    emi = b.texture_png((rs.integers(0, 40, (4, 4, 3))).astype(np.uint8))
    tex = b.material("Textured", base_color_texture=col, metallic_roughness_texture=mr, normal_texture=nrm, emissive_texture=emi, ior=1.33)
    light = b.material("Emitter ceiling", base_color=(0, 0, 0), emissive=(1.0, 0.8, 0.6), emissive_strength=12.0)
    P, I = scenes.icosphere(2)
    Pz = P[:, [0, 2, 1]] * [1, 1, -1]     # any consistent object coordinates will do
    N = Pz / np.linalg.norm(Pz, axis=1, keepdims=True)
    UV = np.stack([np.arctan2(Pz[:, 2], Pz[:, 0]) / (2 * math.pi) + 0.5, np.arccos(np.clip(Pz[:, 1], -1, 1)) / math.pi], -1)
    sphere_plain = lambda mat: b.mesh([dict(positions=Pz, indices=I, material=mat, u32=u32)])
    sphere_full = b.mesh([dict(positions=Pz, indices=I, material=tex, normals=N, texcoords=UV, u32=u32)])
    floor = b.mesh([dict(positions=[(-6, 0, -6), (6, 0, -6), (6, 0, 6), (-6, 0, 6)], indices=[0, 2, 1, 0, 3, 2], material=white)])
    lamp = b.mesh([dict(positions=[(-1.5, 0, -1.5), (1.5, 0, -1.5), (1.5, 0, 1.5), (-1.5, 0, 1.5)], indices=[0, 1, 2, 0, 2, 3], material=light)])
    b.node(mesh=floor)
    b.node(mesh=lamp, translation=(0, 5, 0))
    kids = [b.node(mesh=sphere_plain(mirror), translation=(-2.2, 0, 0), root=False),
            b.node(mesh=sphere_plain(glass), translation=(0, 0, 0), scale=(0.9, 1.1, 0.9), root=False),
            b.node(mesh=sphere_plain(gold), translation=(2.2, 0, 0), rotation=(0, math.sin(0.3), 0, math.cos(0.3)), root=False),
            b.node(mesh=sphere_full, translation=(0, 0, 2.4), rotation=(math.sin(0.2), 0, 0, math.cos(0.2)), root=False)]
    b.node(children=kids, translation=(0, 1.0, 0), rotation=(0, math.sin(0.1), 0, math.cos(0.1)))
    b.node(camera=b.camera(0.7), matrix=assets.look_at_yup((0.5, 3.0, 8.0), (0, 1.0, 0)))
    open(path_glb, "wb").write(b.tobytes())
    open(path_exr, "wb").write(assets.exr_bytes(scenes.sky_sun_equirect(64, 32), "RGB", "half", "zip"))


def write_cornell(path_glb, path_exr):
    """BASELINE.json configs[1]: Cornell box with an emissive ceiling quad ("Emitter…": sampled), black environment.
    Y-up glTF coordinates; five Lambert walls, a short and a tall box, the light just below the ceiling."""
    b = assets.GlbBuilder()
    white = b.material("White", base_color=(0.73, 0.73, 0.73), metallic=0.0, roughness=1.0)
    red = b.material("Red", base_color=(0.65, 0.05, 0.05), metallic=0.0, roughness=1.0)
    green = b.material("Green", base_color=(0.12, 0.45, 0.15), metallic=0.0, roughness=1.0)
    light = b.material("Emitter", base_color=(0, 0, 0), emissive=(1.0, 0.85, 0.6), emissive_strength=15.0)

    def quad(p0, p1, p2, p3, mat):
        return b.mesh([dict(positions=[p0, p1, p2, p3], indices=[0, 1, 2, 0, 2, 3], material=mat)])

    def box(lo, hi, mat):
        (x0, y0, z0), (x1, y1, z1) = lo, hi
        P = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
        I = [0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7, 0, 1, 5, 0, 5, 4, 3, 6, 2, 3, 7, 6, 0, 4, 7, 0, 7, 3, 1, 2, 6, 1, 6, 5]
        return b.mesh([dict(positions=P, indices=I, material=mat)])

    b.node(mesh=quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), white))        # floor
    b.node(mesh=quad((-1, 2, -1), (-1, 2, 1), (1, 2, 1), (1, 2, -1), white))        # ceiling
    b.node(mesh=quad((-1, 0, -1), (-1, 2, -1), (1, 2, -1), (1, 0, -1), white))      # back wall
    b.node(mesh=quad((-1, 0, -1), (-1, 0, 1), (-1, 2, 1), (-1, 2, -1), red))        # left
    b.node(mesh=quad((1, 0, -1), (1, 2, -1), (1, 2, 1), (1, 0, 1), green))          # right
    b.node(mesh=quad((-0.3, 1.98, -0.3), (0.3, 1.98, -0.3), (0.3, 1.98, 0.3), (-0.3, 1.98, 0.3), light))
    b.node(mesh=box((-0.3, 0, -0.3), (0.3, 0.6, 0.3), white), translation=(0.35, 0, 0.3), rotation=(0, math.sin(-0.15), 0, math.cos(-0.15)))
    b.node(mesh=box((-0.3, 0, -0.3), (0.3, 1.2, 0.3), white), translation=(-0.35, 0, -0.3), rotation=(0, math.sin(0.2), 0, math.cos(0.2)))
    b.node(camera=b.camera(0.69), matrix=assets.look_at_yup((0, 1.0, 3.9), (0, 1.0, 0)))
    open(path_glb, "wb").write(b.tobytes())
    open(path_exr, "wb").write(assets.exr_bytes(np.zeros((1, 1, 4), np.float32)))


def write_bathroom_standin(path_glb, path_exr, spheres=48, order=5, tex=64, env=(2048, 1024)):
    """Stand-in for BASELINE.json configs[2]/[3] ("Salle de bain": the asset is not in the reference tree nor on this machine):
    about a million TEXTURED triangles with everything the importer handles on real assets (World.zig:44-363) — `spheres`
    order-`order` icospheres (20 480 triangles each) with per-vertex normals and texcoords, each with its own PNG base colour,
    metallic-roughness and normal map (3 x spheres textures + the room's), a three-level node hierarchy (room -> shelf -> object)
    with TRS at every level (uniform scales only: under a NON-uniform scale the reference's Frame::inSpace, reflection_frame.hlsl:24-30,
    yields non-orthogonal shading frames, hence non-unit bounce directions and, about once in 3e6 samples, a NaN from the environment
    lookup's sqrt(1 - |z|), mappings.hlsl:87 — faithfully reproduced by the oracle and the HIP path alike, but useless in a fixture), glass (KHR_materials_transmission + ior), an emissive-strength "Emitter" panel
    (sampled) and an emissive TEXTURE, tiled room walls, and a 2048x1024 HDR environment stored as a PIZ-compressed HALF EXR."""
    rs = np.random.default_rng(2024)
    b = assets.GlbBuilder(interleaved=True)

    def noise_tex(base, amp, size=tex, normal=False):
        yy, xx = np.mgrid[0:size, 0:size].astype(np.float64) / size
        f = sum(np.sin(2 * math.pi * (k * xx * rs.integers(1, 4) + k * yy * rs.integers(1, 4) + rs.random())) / k for k in (1, 2, 4))[..., None]
        if normal:     # tangent-space normal map: x, y around 0.5, z near 1 (only RG are read, World.zig:57-62)
            img = np.concatenate([0.5 + 0.12 * f, 0.5 + 0.12 * np.roll(f, size // 3, 0), np.ones_like(f)], -1)
        else:
            img = np.asarray(base)[None, None, :] + amp * f * np.asarray([1.0, 0.8, 0.6])
        return (np.clip(img, 0, 1) * 255).astype(np.uint8)

    P, I = scenes.icosphere(order)
    Py = P[:, [0, 2, 1]] * [1, 1, -1]
    N = Py / np.linalg.norm(Py, axis=1, keepdims=True)
    UV = np.stack([np.arctan2(Py[:, 2], Py[:, 0]) / (2 * math.pi) + 0.5, np.arccos(np.clip(Py[:, 1], -1, 1)) / math.pi], -1) * [4.0, 2.0]   # tiles: repeat addressing
    glass = b.material("Glass", transmission=1.0, ior=1.52)
    chrome = b.material("Chrome", metallic=1.0, roughness=0.0)
    objects = []
    for k in range(spheres):
        if k % 8 == 6:
            mat = glass
        elif k % 8 == 7:
            mat = chrome
        else:
            col = b.texture_png(noise_tex(rs.uniform(0.2, 0.9, 3), 0.25))
            mr = b.texture_png(noise_tex((0.5 * (k % 3 == 0), rs.uniform(0.15, 0.9), 0.0), 0.2))          # R = metalness, G = roughness (World.zig:171-174)
            nm = b.texture_png(noise_tex(None, None, normal=True))
            mat = b.material("Ceramic %d" % k, base_color_texture=col, metallic_roughness_texture=mr, normal_texture=nm, ior=1.45 + 0.01 * (k % 5))
        objects.append(b.mesh([dict(positions=Py, indices=I, material=mat, normals=N, texcoords=UV, u32=(k % 2 == 0))]))
    # the room (Y up): tiled floor and walls, a ceiling light panel, a glowing sign (emissive texture)
    tile = b.material("Tiles", base_color_texture=b.texture_png(noise_tex((0.8, 0.85, 0.9), 0.1)), metallic_roughness_texture=b.texture_png(noise_tex((0.0, 0.35, 0.0), 0.15)),
                      normal_texture=b.texture_png(noise_tex(None, None, normal=True)))
    W, H, D = 7.0, 3.2, 5.0

    def quad(p0, p1, p2, p3, mat, rep):
        n = np.cross(np.subtract(p1, p0), np.subtract(p3, p0)); n = n / np.linalg.norm(n)
        return b.mesh([dict(positions=[p0, p1, p2, p3], indices=[0, 1, 2, 0, 2, 3], material=mat, normals=[n] * 4, texcoords=[(0, 0), (rep, 0), (rep, rep), (0, rep)])])
    room = [b.node(mesh=quad((-W, 0, D), (W, 0, D), (W, 0, -D), (-W, 0, -D), tile, 12), root=False),
            b.node(mesh=quad((-W, 0, -D), (W, 0, -D), (W, H, -D), (-W, H, -D), tile, 8), root=False),
            b.node(mesh=quad((-W, 0, D), (-W, 0, -D), (-W, H, -D), (-W, H, D), tile, 8), root=False),
            b.node(mesh=quad((W, 0, -D), (W, 0, D), (W, H, D), (W, H, -D), tile, 8), root=False)]
    lamp = b.material("Emitter panel", base_color=(0, 0, 0), emissive=(1.0, 0.95, 0.85), emissive_strength=18.0)
    room.append(b.node(mesh=quad((-1.5, H - 0.01, -1.0), (1.5, H - 0.01, -1.0), (1.5, H - 0.01, 1.0), (-1.5, H - 0.01, 1.0), lamp, 1), root=False))
    sign = b.material("Sign", base_color=(0.02, 0.02, 0.02), emissive_texture=b.texture_png(noise_tex((0.9, 0.2, 0.1), 0.5, 32)), emissive=(1, 1, 1))
    room.append(b.node(mesh=quad((-2, 1.6, -D + 0.02), (2, 1.6, -D + 0.02), (2, 2.4, -D + 0.02), (-2, 2.4, -D + 0.02), sign, 1), root=False))
    # shelves: room -> shelf (rotated, non-uniformly scaled) -> object (own TRS)
    per = 8
    for sidx in range((spheres + per - 1) // per):
        kids = []
        for j, m in enumerate(objects[sidx * per:(sidx + 1) * per]):
            r = 0.28 + 0.05 * ((sidx + j) % 3)
            kids.append(b.node(mesh=m, translation=((j - (per - 1) / 2) * 0.9, r, 0.15 * ((j % 2) * 2 - 1)), scale=(r, r, r),
                               rotation=(0, math.sin(0.3 * j), 0, math.cos(0.3 * j)), root=False))
        ang = 0.12 * (sidx - 2.5)
        room.append(b.node(children=kids, translation=(0.0, 0.02 + 0.01 * sidx, -D + 1.0 + 1.3 * sidx), rotation=(0, math.sin(ang / 2), 0, math.cos(ang / 2)),
                           scale=(1.0 + 0.05 * sidx,) * 3, root=False))
    b.node(children=room, translation=(0.0, -0.5, 0.0))
    b.node(camera=b.camera(0.75, 16.0 / 9.0), matrix=assets.look_at_yup((0.0, 2.3, 9.5), (0.0, 0.6, 0.0)))
    open(path_glb, "wb").write(b.tobytes())
    sky = scenes.sky_sun_equirect(*env)
    open(path_exr, "wb").write(assets.exr_bytes(sky, "RGB", "half", "piz"))
    return dict(triangles=spheres * len(I) + 2 * 6, textures=len(b.j["textures"]), nodes=len(b.j["nodes"]))


def write_random_glb(path_glb, path_exr, seed):
    """a glTF file drawn from a seed, over everything World.zig:44-349 and Camera.zig:26-51 branch on: materials with every combination of base-colour / metallic-roughness /
    normal / emissive textures and factors — the exact (0, 1) and (1, 0) factor pairs, transmission 1 and below 1, ior, emissive strength, names with and without the
    "Emitter" prefix, materials without a pbrMetallicRoughness block —, meshes of one to three primitives with and without normals / texcoords (u16 and u32 indices,
    separate and interleaved buffers), node trees up to four deep with TRS in every subset, matrices, empty nodes, meshes used by several nodes, the camera anywhere in
    the tree (a second camera after it must be ignored)"""
    rs = np.random.default_rng(seed)
    b = assets.GlbBuilder(interleaved=bool(rs.random() < 0.3))
    u32 = bool(rs.random() < 0.3)

    def png(lo, hi, c=3):
        w, h = int(rs.integers(1, 9)), int(rs.integers(1, 9))
        return b.texture_png(rs.integers(lo, hi, (h, w, 3), dtype=np.uint8))
    mats = []
    for i in range(int(rs.integers(2, 7))):
        kw = {}
        r = rs.random()
        if r < 0.2: kw.update(metallic=0.0, roughness=1.0)
        elif r < 0.35: kw.update(metallic=1.0, roughness=0.0)
        elif r < 0.45: kw.update(metallic=0.0, roughness=0.0)
        else: kw.update(metallic=float(rs.random()), roughness=float(rs.uniform(0.05, 1.0)))
        if rs.random() < 0.2: kw.update(transmission=1.0 if rs.random() < 0.7 else float(rs.uniform(0.1, 0.99)))
        if rs.random() < 0.4: kw.update(ior=float(rs.uniform(1.1, 2.0)))
        if rs.random() < 0.4: kw.update(base_color_texture=png(0, 256))
        else: kw.update(base_color=tuple(rs.random(3)))
        if rs.random() < 0.3: kw.update(metallic_roughness_texture=png(20, 230))
        if rs.random() < 0.3: kw.update(normal_texture=b.texture_png(np.concatenate([128 + rs.integers(-30, 30, (4, 4, 2)), np.full((4, 4, 1), 255)], -1).astype(np.uint8)))
        emit = rs.random() < 0.35
        if emit and rs.random() < 0.4: kw.update(emissive_texture=png(0, 60))
        elif emit: kw.update(emissive=tuple(rs.random(3)), **({"emissive_strength": float(rs.uniform(0.5, 20.0))} if rs.random() < 0.6 else {}))
        name = ("Emitter %d" % i) if (emit and rs.random() < 0.6) else ("Emit%d" % i if rs.random() < 0.1 else "Mat %d" % i)
        mats.append(b.material(name, **kw))
    if rs.random() < 0.3:      # a material that names nothing but itself: every default of the glTF schema
        b.j["materials"].append({"name": "Bare"}); mats.append(len(b.j["materials"]) - 1)

    def prim():
        kind = int(rs.integers(0, 3))
        if kind == 0:
            P, I = scenes.icosphere(int(rs.integers(0, 2))); P = P * rs.uniform(0.3, 1.0, (1, 3))
        elif kind == 1:
            e = float(rs.uniform(0.5, 2.5)); P = np.array([(-e, 0, -e), (e, 0, -e), (e, 0, e), (-e, 0, e)], np.float32); I = np.array([0, 2, 1, 0, 3, 2])
        else:
            n = int(rs.integers(1, 20)); P = rs.normal(size=(3 * n, 3)) * rs.uniform(0.2, 1.0); I = np.arange(3 * n)
        d = dict(positions=np.asarray(P, np.float32), indices=np.asarray(I).reshape(-1), material=mats[int(rs.integers(len(mats)))], u32=u32)
        both = rs.random() < 0.4
        if both or rs.random() < 0.3:
            N = rs.normal(size=(len(P), 3)); d["normals"] = (N / np.linalg.norm(N, axis=1, keepdims=True)).astype(np.float32)
        if both or rs.random() < 0.3:
            d["texcoords"] = (rs.random((len(P), 2)) * rs.uniform(0.5, 3.0) - 0.5).astype(np.float32)
        return d
    meshes = [b.mesh([prim() for _ in range(int(rs.integers(1, 4)))]) for _ in range(int(rs.integers(1, 5)))]

    def xform():
        kw = {}
        r = rs.random()
        if r < 0.25:
            A = np.eye(4); A[:3, :3] = rs.normal(size=(3, 3)) * rs.uniform(0.4, 1.2); A[:3, 3] = rs.normal(size=3) * 2.0
            kw["matrix"] = A
        elif r < 0.9:
            if rs.random() < 0.7: kw["translation"] = tuple(rs.normal(size=3) * 2.0)
            if rs.random() < 0.6:
                q = rs.normal(size=4); kw["rotation"] = tuple(q / np.linalg.norm(q))
            if rs.random() < 0.4: kw["scale"] = tuple(rs.uniform(0.4, 1.6, 3) * rs.choice([-1.0, 1.0], 3, p=[0.15, 0.85]))
        return kw
    cam_at = int(rs.integers(0, 6))
    placed = [0]

    def camera_node(root):
        q = rs.normal(size=4) * 0.15 + np.array([0, 0, 0, 1.0])
        return b.node(camera=b.camera(float(rs.uniform(0.4, 1.1))), translation=(float(rs.normal() * 0.5), float(rs.normal() * 0.5 + 1.0), float(rs.uniform(5.0, 9.0))), rotation=tuple(q / np.linalg.norm(q)), root=root)

    def tree(depth, root):
        kids = []
        if depth < 3:
            kids = [tree(depth + 1, False) for _ in range(int(rs.integers(0, 3 if depth else 4)))]
        if placed[0] == cam_at:
            kids.append(camera_node(False))
        placed[0] += 1
        return b.node(mesh=meshes[int(rs.integers(len(meshes)))] if rs.random() < 0.75 else None, children=kids or None, root=root, **xform())
    for _ in range(int(rs.integers(1, 4))):
        tree(0, True)
    camera_node(True)    # (the file's first camera node when none was placed in a tree; otherwise the one to ignore)
    open(path_glb, "wb").write(b.tobytes())
    if rs.random() < 0.5:
        open(path_exr, "wb").write(assets.exr_bytes(np.array([[[*(rs.random(3) * 0.8), 1.0]]], np.float32)))
    else:
        open(path_exr, "wb").write(assets.exr_bytes(scenes.sky_sun_equirect(32, 16), "RGB", "half", "zip"))
