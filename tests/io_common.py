"""Shared helpers for the scene-I/O tests: asset generation (moonshine_amd.assets) and the importer→oracle shim."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

from moonshine_amd import assets, scenes

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
