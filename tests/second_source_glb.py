"""A second statement of the reference's glTF import rules, in Python, straight from the reference's own text — test infrastructure.

The product imports a .glb with moonshine_amd/host/glb.cpp; the GPU-side tests so far fed the ORACLE through that same importer (tests/shim/glb_oracle_shim.cpp), which
proves "same importer -> same film", not that the importer follows the reference.  This module reads the GLB container and its JSON with Python's own tools and
builds the scene DIRECTLY through the scene API (any context with create_mesh / create_texture / create_material / create_instance / create_lens: the product's
ctypes binding or the oracle's), applying, line by line:

  World.zig:44-228   gltfMaterialToMaterial — normal texture (PNG RGB -> R,G as r8g8_unorm, else the 1x1 default (0.5, 0.5)); emissive (PNG -> RGBA8 sRGB with
                     alpha 255, else emissive_factor * emissive_strength); transmission_factor == 1 -> Glass{ior}; base colour (PNG -> RGBA8 sRGB, else the factor's
                     RGB); metallicRoughness texture -> R = metalness, G = roughness as two r8_unorm textures, StandardPBR; without it (metallic, roughness) ==
                     (0, 1) -> Lambert, (1, 0) -> PerfectMirror, else StandardPBR with 1x1 constants; ior from KHR_materials_ior (default 1.5)
  World.zig:254-349  fromGlb — every NODE with a mesh: one new mesh per primitive (positions, optional texcoords / normals, indices widened to u32), one
                     Geometry{mesh, primitive.material, sampled = material name starts with "Emitter"} per primitive, one instance with the node's GLOBAL transform,
                     rows (x, z, y) of the glTF matrix (:341-345)
  Camera.zig:26-51   Lens.fromGlb — first node with a camera; the same row swap; origin = T.mul_point(0), forward = T.mul_vec((0, 0, -1)).unit(), up = T.mul_vec((0, 1, 0)),
                     vfov = yfov, aperture 0, focus distance 1 (vector.zig: dot products summed left to right, unit() = division by the length)

Node transforms: zgltf's getGlobalTransform (the dependency is not vendored in the reference tree) — local matrix = the node's `matrix`, or T * R * S from the
quaternion (the standard rotation-matrix formula, columns scaled), parents multiplied on from the left up the chain.  All arithmetic in float32 (numpy scalars: one
IEEE operation per operator, like the C++ built with -ffp-contract=off), in the order written here."""
import io
import json
import struct

import numpy as np

F = np.float32
GLASS, LAMBERT, PERFECT_MIRROR, STANDARD_PBR = 0, 1, 2, 3     # world.hlsl:31-36 (the scene API's material types)


def _chunks(data):
    magic, version, total = struct.unpack_from("<4sII", data, 0)
    assert magic == b"glTF" and version == 2
    pos, js, bin_ = 12, None, None
    while pos + 8 <= total:
        n, t = struct.unpack_from("<II", data, pos)
        body = data[pos + 8:pos + 8 + n]
        if t == 0x4E4F534A:
            js = json.loads(body.decode())
        elif t == 0x004E4942 and bin_ is None:
            bin_ = body
        pos += 8 + n
    return js, bin_


def _accessor(j, bin_, idx):
    a = j["accessors"][idx]
    bv = j["bufferViews"][a["bufferView"]]
    comps = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}[a["type"]]
    dt = {5126: "<f4", 5125: "<u4", 5123: "<u2", 5121: "u1"}[a["componentType"]]
    size = np.dtype(dt).itemsize * comps
    stride = bv.get("byteStride", 0) or size
    base = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    out = np.zeros((a["count"], comps), dt)
    for i in range(a["count"]):
        out[i] = np.frombuffer(bin_, dt, comps, base + i * stride)
    return out


def _png_rgb(j, bin_, texture_index):
    from PIL import Image
    img = j["images"][j["textures"][texture_index]["source"]]
    assert img["mimeType"] == "image/png"                                   # World.zig:50
    bv = j["bufferViews"][img["bufferView"]]
    im = Image.open(io.BytesIO(bin_[bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]))
    assert im.mode == "RGB", "the reference reads img.pixels.rgb24"
    return np.asarray(im, np.uint8)


def _local(node):
    """column-major 4x4 like zgltf's: m[col][row]"""
    m = [[F(0)] * 4 for _ in range(4)]
    if "matrix" in node:
        for i, v in enumerate(node["matrix"]):
            m[i // 4][i % 4] = F(v)
        return m
    t = [F(v) for v in node.get("translation", (0, 0, 0))]
    x, y, z, w = [F(v) for v in node.get("rotation", (0, 0, 0, 1))]
    s = [F(v) for v in node.get("scale", (1, 1, 1))]
    one, two = F(1), F(2)
    m[0][0] = (one - two * (y * y + z * z)) * s[0]; m[0][1] = (two * (x * y + z * w)) * s[0]; m[0][2] = (two * (x * z - y * w)) * s[0]
    m[1][0] = (two * (x * y - z * w)) * s[1]; m[1][1] = (one - two * (x * x + z * z)) * s[1]; m[1][2] = (two * (y * z + x * w)) * s[1]
    m[2][0] = (two * (x * z + y * w)) * s[2]; m[2][1] = (two * (y * z - x * w)) * s[2]; m[2][2] = (one - two * (x * x + y * y)) * s[2]
    m[3][0], m[3][1], m[3][2], m[3][3] = t[0], t[1], t[2], one
    return m


def _mul(a, b):
    r = [[F(0)] * 4 for _ in range(4)]
    for c in range(4):
        for rr in range(4):
            s = F(0)
            for k in range(4):
                s = s + a[k][rr] * b[c][k]
            r[c][rr] = s
    return r


def _global(j, i):
    parent = {}
    for n, node in enumerate(j["nodes"]):
        for c in node.get("children", ()):
            parent[c] = n
    m = _local(j["nodes"][i])
    p = parent.get(i)
    while p is not None:
        m = _mul(_local(j["nodes"][p]), m)
        p = parent.get(p)
    return m


def _z_up(m):
    """World.zig:341-345 / Camera.zig:36-40: rows (x, z, y) of the glTF matrix"""
    return np.array([[m[0][0], m[1][0], m[2][0], m[3][0]], [m[0][2], m[1][2], m[2][2], m[3][2]], [m[0][1], m[1][1], m[2][1], m[3][1]]], np.float32)


def _material(ctx, j, bin_, m):
    ext = m.get("extensions", {})
    if "normalTexture" in m:                                                               # World.zig:47-75
        rgb = _png_rgb(j, bin_, m["normalTexture"]["index"])
        normal = ctx.create_texture(np.ascontiguousarray(rgb[..., :2]), rgb.shape[1], rgb.shape[0], "r8g8_unorm")
    else:
        normal = ctx.solid_texture(0.5, 0.5)

    def rgba_srgb(tex):
        rgb = _png_rgb(j, bin_, tex["index"])
        px = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], -1)
        return ctx.create_texture(np.ascontiguousarray(px), rgb.shape[1], rgb.shape[0], "r8g8b8a8_srgb")
    if "emissiveTexture" in m:                                                              # :77-110
        emissive = rgba_srgb(m["emissiveTexture"])
    else:
        ef = [F(v) for v in m.get("emissiveFactor", (0, 0, 0))]
        strength = F(ext.get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0))
        emissive = ctx.solid_texture(float(ef[0] * strength), float(ef[1] * strength), float(ef[2] * strength))
    ior = float(F(ext.get("KHR_materials_ior", {}).get("ior", 1.5)))
    if F(ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0)) == F(1.0):   # :118-121
        return ctx.create_material(GLASS, normal, emissive, ior=ior)
    pbr = m.get("pbrMetallicRoughness", {})
    if "baseColorTexture" in pbr:                                                           # :123-157
        color = rgba_srgb(pbr["baseColorTexture"])
    else:
        bc = [F(v) for v in pbr.get("baseColorFactor", (1, 1, 1, 1))]
        color = ctx.solid_texture(float(bc[0]), float(bc[1]), float(bc[2]))
    metallic, roughness = F(pbr.get("metallicFactor", 1.0)), F(pbr.get("roughnessFactor", 1.0))
    if "metallicRoughnessTexture" in pbr:                                                   # :159-204: R = metalness, G = roughness
        rgb = _png_rgb(j, bin_, pbr["metallicRoughnessTexture"]["index"])
        metal = ctx.create_texture(np.ascontiguousarray(rgb[..., 0]), rgb.shape[1], rgb.shape[0], "r8_unorm")
        rough = ctx.create_texture(np.ascontiguousarray(rgb[..., 1]), rgb.shape[1], rgb.shape[0], "r8_unorm")
        return ctx.create_material(STANDARD_PBR, normal, emissive, color=color, metalness=metal, roughness=rough, ior=ior)
    if metallic == F(0.0) and roughness == F(1.0):                                          # :206-212
        return ctx.create_material(LAMBERT, normal, emissive, color=color, ior=ior)
    if metallic == F(1.0) and roughness == F(0.0):                                          # :213-216
        return ctx.create_material(PERFECT_MIRROR, normal, emissive, ior=ior)
    return ctx.create_material(STANDARD_PBR, normal, emissive, color=color, metalness=ctx.solid_texture(float(metallic)), roughness=ctx.solid_texture(float(roughness)), ior=ior)


def load(ctx, path):
    """-> lens handle; the scene is built on `ctx`"""
    j, bin_ = _chunks(open(path, "rb").read())
    mats = [_material(ctx, j, bin_, m) for m in j.get("materials", ())]                      # World.zig:234-248: every material, in file order
    for ni, node in enumerate(j["nodes"]):                                                   # :262-349
        if "mesh" not in node:
            continue
        geos = []
        for pr in j["meshes"][node["mesh"]]["primitives"]:
            at = pr["attributes"]
            assert set(at) <= {"POSITION", "NORMAL", "TEXCOORD_0"}, "error.UnhandledAttribute (World.zig:308-311)"
            pos = _accessor(j, bin_, at["POSITION"]).astype(np.float32)
            nrm = _accessor(j, bin_, at["NORMAL"]).astype(np.float32) if "NORMAL" in at else None
            uv = _accessor(j, bin_, at["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in at else None
            idx = _accessor(j, bin_, pr["indices"]).astype(np.uint32).reshape(-1, 3)     # (u16 in the reference; wider index types are this project's extension)
            mesh = ctx.create_mesh(pos, idx, normals=nrm, texcoords=uv)
            name = j["materials"][pr["material"]].get("name", "")
            geos.append((mesh, mats[pr["material"]], name.startswith("Emitter")))           # :266-270
        ctx.create_instance(geos, transform=_z_up(_global(j, ni)))
    for ni, node in enumerate(j["nodes"]):                                                   # Camera.zig:28-30: the first node with a camera
        if "camera" not in node:
            continue
        T = _z_up(_global(j, ni))

        def dot4(r, v):                                                                      # vector.zig:192: x*x' + y*y' + z*z' + w*w', left to right
            return r[0] * v[0] + r[1] * v[1] + r[2] * v[2] + r[3] * v[3]
        point = lambda v: [dot4(T[k], (F(v[0]), F(v[1]), F(v[2]), F(1))) for k in range(3)]
        vec = lambda v: [dot4(T[k], (F(v[0]), F(v[1]), F(v[2]), F(0))) for k in range(3)]
        origin, f, up = point((0, 0, 0)), vec((0, 0, -1)), vec((0, 1, 0))
        length = np.sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2])
        forward = [f[0] / length, f[1] / length, f[2] / length]
        yfov = float(F(j["cameras"][node["camera"]]["perspective"]["yfov"]))
        return ctx.create_lens(ctx.make_lens(tuple(float(v) for v in origin), tuple(float(v) for v in forward), tuple(float(v) for v in up), yfov, 0.0, 1.0))
    raise RuntimeError("error.NoCameraInGlb")
