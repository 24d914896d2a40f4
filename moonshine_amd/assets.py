"""Writers for the file formats on either side of the hot path, used to generate test/bench assets offline (there is
no network, and the reference ships no assets): binary glTF 2.0 (.glb), PNG (8-bit RGB) and scanline OpenEXR
(FLOAT or HALF, NONE / ZIPS / ZIP).  Pure Python + numpy + zlib; independent of the C++ codecs they are used to test."""
import json
import struct
import zlib

import numpy as np


def png_rgb8(img):
    """(H, W, 3) uint8 -> PNG bytes (filter type 0 + one Paeth/Sub row to exercise the unfilter)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, _ = img.shape
    raw = bytearray()
    for y in range(h):
        row = img[y].reshape(-1).astype(np.int16)
        if y % 3 == 1:      # Sub filter
            prev = np.concatenate([np.zeros(3, np.int16), row[:-3]])
            raw += b"\x01" + ((row - prev) & 0xFF).astype(np.uint8).tobytes()
        elif y % 3 == 2:    # Up filter
            up = img[y - 1].reshape(-1).astype(np.int16)
            raw += b"\x02" + ((row - up) & 0xFF).astype(np.uint8).tobytes()
        else:
            raw += b"\x00" + row.astype(np.uint8).tobytes()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")


class GlbBuilder:
    """Minimal glTF 2.0 writer: meshes with POSITION/NORMAL/TEXCOORD_0 + indices, materials, PNG textures, one camera."""

    def __init__(self, interleaved=False):
        self.interleaved = interleaved
        self.bin = bytearray()
        self.j = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": []}], "nodes": [], "meshes": [], "materials": [],
                  "accessors": [], "bufferViews": [], "buffers": [{}], "cameras": [], "textures": [], "images": []}

    def _view(self, data, target=None):
        while len(self.bin) % 4:
            self.bin += b"\0"
        off = len(self.bin)
        self.bin += data
        bv = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if target:
            bv["target"] = target
        self.j["bufferViews"].append(bv)
        return len(self.j["bufferViews"]) - 1

    def _accessor(self, arr, ctype, typ, target=None):
        arr = np.ascontiguousarray(arr)
        acc = {"bufferView": self._view(arr.tobytes(), target), "componentType": ctype, "count": int(arr.shape[0]), "type": typ}
        if typ == "VEC3" and ctype == 5126:
            acc["min"] = [float(x) for x in arr.min(0)]; acc["max"] = [float(x) for x in arr.max(0)]
        self.j["accessors"].append(acc)
        return len(self.j["accessors"]) - 1

    def texture_png(self, rgb8):
        self.j["images"].append({"bufferView": self._view(png_rgb8(rgb8)), "mimeType": "image/png"})
        self.j["textures"].append({"source": len(self.j["images"]) - 1})
        return len(self.j["textures"]) - 1

    def material(self, name, base_color=(1, 1, 1), metallic=0.0, roughness=1.0, emissive=(0, 0, 0), emissive_strength=None, ior=None,
                 transmission=None, base_color_texture=None, metallic_roughness_texture=None, normal_texture=None, emissive_texture=None):
        m = {"name": name, "pbrMetallicRoughness": {"baseColorFactor": list(map(float, base_color)) + [1.0], "metallicFactor": float(metallic), "roughnessFactor": float(roughness)},
             "emissiveFactor": list(map(float, emissive))}
        if base_color_texture is not None:
            m["pbrMetallicRoughness"]["baseColorTexture"] = {"index": base_color_texture}
        if metallic_roughness_texture is not None:
            m["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": metallic_roughness_texture}
        if normal_texture is not None:
            m["normalTexture"] = {"index": normal_texture}
        if emissive_texture is not None:
            m["emissiveTexture"] = {"index": emissive_texture}
        ext = {}
        if emissive_strength is not None:
            ext["KHR_materials_emissive_strength"] = {"emissiveStrength": float(emissive_strength)}
        if ior is not None:
            ext["KHR_materials_ior"] = {"ior": float(ior)}
        if transmission is not None:
            ext["KHR_materials_transmission"] = {"transmissionFactor": float(transmission)}
        if ext:
            m["extensions"] = ext
        self.j["materials"].append(m)
        return len(self.j["materials"]) - 1

    def mesh(self, primitives):
        """primitives: list of dict(positions, indices, material, normals=None, texcoords=None); u16 indices when they fit."""
        prims = []
        for p in primitives:
            pos = np.asarray(p["positions"], np.float32)
            idx = np.asarray(p["indices"]).reshape(-1)
            small = pos.shape[0] <= 65535 and not p.get("u32")
            if self.interleaved and p.get("normals") is not None and p.get("texcoords") is not None:
                # one vertex buffer, attributes interleaved (bufferView.byteStride = 32, accessor.byteOffset 0 / 12 / 24): what exporters write
                inter = np.concatenate([pos, np.asarray(p["normals"], np.float32), np.asarray(p["texcoords"], np.float32)], 1).astype(np.float32)
                bv = self._view(inter.tobytes(), 34962)
                self.j["bufferViews"][bv]["byteStride"] = 32
                attrs = {}
                for name, off, typ, arr in (("POSITION", 0, "VEC3", pos), ("NORMAL", 12, "VEC3", None), ("TEXCOORD_0", 24, "VEC2", None)):
                    acc = {"bufferView": bv, "byteOffset": off, "componentType": 5126, "count": int(pos.shape[0]), "type": typ}
                    if arr is not None:
                        acc["min"] = [float(x) for x in arr.min(0)]; acc["max"] = [float(x) for x in arr.max(0)]
                    self.j["accessors"].append(acc); attrs[name] = len(self.j["accessors"]) - 1
                prims.append({"attributes": attrs, "material": p["material"], "mode": 4,
                              "indices": self._accessor(idx.astype(np.uint16 if small else np.uint32), 5123 if small else 5125, "SCALAR", 34963)})
                continue
            attrs = {"POSITION": self._accessor(pos, 5126, "VEC3", 34962)}
            if p.get("normals") is not None:
                attrs["NORMAL"] = self._accessor(np.asarray(p["normals"], np.float32), 5126, "VEC3", 34962)
            if p.get("texcoords") is not None:
                attrs["TEXCOORD_0"] = self._accessor(np.asarray(p["texcoords"], np.float32), 5126, "VEC2", 34962)
            prims.append({"attributes": attrs, "material": p["material"], "mode": 4,
                          "indices": self._accessor(idx.astype(np.uint16 if small else np.uint32), 5123 if small else 5125, "SCALAR", 34963)})
        self.j["meshes"].append({"primitives": prims})
        return len(self.j["meshes"]) - 1

    def node(self, mesh=None, camera=None, translation=None, rotation=None, scale=None, matrix=None, children=None, root=True):
        n = {}
        if mesh is not None:
            n["mesh"] = mesh
        if camera is not None:
            n["camera"] = camera
        if translation is not None:
            n["translation"] = list(map(float, translation))
        if rotation is not None:
            n["rotation"] = list(map(float, rotation))
        if scale is not None:
            n["scale"] = list(map(float, scale))
        if matrix is not None:
            n["matrix"] = [float(x) for x in np.asarray(matrix, np.float64).T.reshape(-1)]   # column-major
        if children:
            n["children"] = list(children)
        self.j["nodes"].append(n)
        i = len(self.j["nodes"]) - 1
        if root:
            self.j["scenes"][0]["nodes"].append(i)
        return i

    def camera(self, yfov, aspect=1.0):
        self.j["cameras"].append({"type": "perspective", "perspective": {"yfov": float(yfov), "aspectRatio": float(aspect), "znear": 0.01}})
        return len(self.j["cameras"]) - 1

    def tobytes(self):
        j = {k: v for k, v in self.j.items() if v or k in ("asset", "scene")}
        while len(self.bin) % 4:
            self.bin += b"\0"
        j["buffers"] = [{"byteLength": len(self.bin)}]
        js = json.dumps(j, separators=(",", ":")).encode()
        js += b" " * (-len(js) % 4)
        total = 12 + 8 + len(js) + 8 + len(self.bin)
        return b"glTF" + struct.pack("<II", 2, total) + struct.pack("<I", len(js)) + b"JSON" + js + struct.pack("<I", len(self.bin)) + b"BIN\0" + bytes(self.bin)


def look_at_yup(eye, target, up=(0, 1, 0)):
    """4x4 camera-to-world matrix of a glTF camera (looks down -z, +y up)."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    f = target - eye; f /= np.linalg.norm(f)
    s = np.cross(f, up); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = s, u, -f, eye
    return m


def exr_bytes(rgba, channels="RGB", pixel_type="float", compression="none"):
    """(H, W, 4) float32 -> scanline OpenEXR bytes.  channels: subset of 'ABGR' letters; pixel_type float|half;
    compression none|zips|zip.  Written from the OpenEXR file-layout document, independently of the C++ reader."""
    a = np.asarray(rgba, np.float32)
    h, w, _ = a.shape
    names = sorted(channels)
    ptype = {"float": 2, "half": 1}[pixel_type]
    comp = {"none": 0, "zips": 2, "zip": 3}[compression]

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", ptype, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp])) \
        + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") \
        + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) \
        + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = {0: 1, 2: 1, 3: 16}[comp]
    col = {"R": 0, "G": 1, "B": 2, "A": 3}
    blocks = []
    for y0 in range(0, h, lines):
        raw = bytearray()
        for y in range(y0, min(y0 + lines, h)):
            for n in names:
                v = a[y, :, col[n]]
                raw += (v.astype(np.float16) if ptype == 1 else v).tobytes()
        data = bytes(raw)
        if comp:
            b = np.frombuffer(data, np.uint8)
            half = (len(b) + 1) // 2
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int16)
            assert len(b[0::2]) == half
            d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
            z = zlib.compress(d.astype(np.uint8).tobytes())
            if len(z) < len(data):
                data = z
        blocks.append((y0, data))
    table = len(hdr)
    off = table + 8 * len(blocks)
    offs, body = [], bytearray()
    for y0, data in blocks:
        offs.append(off + len(body))
        body += struct.pack("<ii", y0, len(data)) + data
    return hdr + b"".join(struct.pack("<Q", o) for o in offs) + bytes(body)


def exr_decode(data):
    """Scanline OpenEXR bytes -> (H, W, 4) float32 (missing channels: RGB 0, A 1).  NONE / ZIPS / ZIP, half / float.
    Pure Python + zlib, written from the file-layout document: the independent check of the C++ writer, and the reader of
    tools/exrdiff.py (runs without the library)."""
    assert struct.unpack_from("<I", data, 0)[0] == 20000630, "not an OpenEXR file"
    pos = 8
    attrs = {}
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
        e = data.index(b"\0", pos); typ = data[pos:e].decode(); pos = e + 1
        n = struct.unpack_from("<I", data, pos)[0]; pos += 4
        attrs[name] = (typ, data[pos:pos + n]); pos += n
    pos += 1
    chans, c, p = [], attrs["channels"][1], 0
    while c[p] != 0:
        e = c.index(b"\0", p); nm = c[p:e].decode(); p = e + 1
        ptype = struct.unpack_from("<i", c, p)[0]; p += 16
        chans.append((nm, ptype))
    comp = attrs["compression"][1][0]
    assert comp in (0, 2, 3), "compression %d not supported by this decoder" % comp
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    lines = 16 if comp == 3 else 1
    nblocks = (h + lines - 1) // lines
    offs = struct.unpack_from("<%dQ" % nblocks, data, pos)
    out = np.zeros((h, w, 4), np.float32); out[..., 3] = 1.0
    col = {"R": 0, "G": 1, "B": 2, "A": 3}
    bpp = {1: 2, 2: 4}
    line_bytes = sum(bpp[t] for _, t in chans) * w
    for o in offs:
        by, n = struct.unpack_from("<ii", data, o)
        rows = min(lines, y1 + 1 - by)
        raw = data[o + 8:o + 8 + n]
        if comp and n < rows * line_bytes:
            d = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int32)
            t = d.copy()
            for i in range(1, len(t)):                      # undo the delta predictor
                t[i] = (t[i - 1] + d[i] - 128) & 255
            half = (len(t) + 1) // 2
            b = np.empty(len(t), np.uint8); b[0::2] = t[:half]; b[1::2] = t[half:]
            raw = b.tobytes()
        p = 0
        for r in range(rows):
            for nm, t in chans:
                v = np.frombuffer(raw, np.float16 if t == 1 else np.float32, w, p).astype(np.float32); p += bpp[t] * w
                if nm in col:
                    out[by - y0 + r, :, col[nm]] = v
    return out
