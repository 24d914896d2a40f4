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


# ---------------- PIZ (OpenEXR's wavelet + Huffman scheme), encoder side ----------------
# Written from the published description of the scheme (OpenEXR "PIZ": per 32-line block the 16-bit words of every channel are
# (1) remapped onto 0..k through a bitmap of the values that occur, (2) transformed by a 2-D two-tap wavelet, level by level,
# (3) Huffman-coded with canonical codes and a run-length escape).  Array-at-a-time numpy here; the C++ reader
# (host/exr.cpp) walks pointers: the two share no code, only the format.
def _piz_wenc(a, b, w14):
    a = a.astype(np.int64); b = b.astype(np.int64)
    if w14:
        sa = np.where(a >= 0x8000, a - 0x10000, a); sb = np.where(b >= 0x8000, b - 0x10000, b)
        return ((sa + sb) >> 1) & 0xffff, (sa - sb) & 0xffff
    ao = (a + 0x8000) & 0xffff
    m = (ao + b) >> 1
    d = ao - b
    m = np.where(d < 0, (m + 0x8000) & 0xffff, m)
    return m, d & 0xffff


def _piz_wavelet(plane, max_value):
    """plane: (ny, nx) array of words, transformed in place level by level (finest first)"""
    a = plane.astype(np.int64)
    ny, nx = a.shape
    w14 = max_value < (1 << 14)
    p, p2 = 1, 2
    while p2 <= min(nx, ny):
        qy, qx = ny // p2, nx // p2
        Y = np.arange(qy) * p2; X = np.arange(qx) * p2
        y0, x0 = np.meshgrid(Y, X, indexing="ij")
        i00, i01 = _piz_wenc(a[y0, x0], a[y0, x0 + p], w14)
        i10, i11 = _piz_wenc(a[y0 + p, x0], a[y0 + p, x0 + p], w14)
        a[y0, x0], a[y0 + p, x0] = _piz_wenc(i00, i10, w14)
        a[y0, x0 + p], a[y0 + p, x0 + p] = _piz_wenc(i01, i11, w14)
        if nx & p:      # a column without a right-hand partner: vertical pairs only
            xl = qx * p2
            a[Y, xl], a[Y + p, xl] = _piz_wenc(a[Y, xl], a[Y + p, xl], w14)
        if ny & p:      # a row without a partner below: horizontal pairs only
            yl = qy * p2
            a[yl, X], a[yl, X + p] = _piz_wenc(a[yl, X], a[yl, X + p], w14)
        p, p2 = p2, p2 * 2
    return a.astype(np.uint16)


def _piz_huffman(words, use_rle=True):
    """-> the Huffman section: 20-byte header {im, iM, table bytes, data bits, 0}, packed code lengths, coded words"""
    import heapq
    words = np.asarray(words, np.int64)
    freq = np.bincount(words, minlength=65537).astype(np.int64)
    im = int(np.flatnonzero(freq)[0]); iM = int(np.flatnonzero(freq)[-1]) + 1
    freq[iM] = 1                                   # the run-length escape takes the first free symbol above the data
    heap = [(int(freq[s]), int(s), (int(s),)) for s in np.flatnonzero(freq)]
    heapq.heapify(heap)
    length = np.zeros(65537, np.int64)
    if len(heap) == 1:
        length[heap[0][1]] = 1
    while len(heap) > 1:
        fa, ka, sa = heapq.heappop(heap); fb, kb, sb = heapq.heappop(heap)
        for t in sa + sb:
            length[t] += 1
        heapq.heappush(heap, (fa + fb, min(ka, kb), sa + sb))
    assert length.max() <= 58
    # canonical codes: the longest codes get the numerically smallest values; within a length, symbols in index order
    count = np.bincount(length, minlength=59)
    first = np.zeros(59, np.int64); c = 0
    for l in range(58, 0, -1):
        first[l] = c; c = (c + int(count[l])) >> 1
    code = np.zeros(65537, object); nxt = [int(x) for x in first]
    for sym in np.flatnonzero(length):
        code[sym] = nxt[length[sym]]; nxt[length[sym]] += 1
    bits = []                                       # (value, number of bits), most significant bit first

    def put(v, n):
        bits.append((int(v), int(n)))
    sym = im
    while sym <= iM:                                # 6 bits per length; 59..62 = 2..5 zeros, 63 + 8 bits = 6..261 zeros
        if length[sym] == 0:
            run = 1
            while sym + run <= iM and length[sym + run] == 0 and run < 261:
                run += 1
            if run >= 6:
                put(63, 6); put(run - 6, 8); sym += run; continue
            if run >= 2:
                put(59 + run - 2, 6); sym += run; continue
        put(length[sym], 6); sym += 1
    table_bits = sum(n for _, n in bits)
    bits.append((0, -table_bits % 8))               # the table ends on a byte boundary
    start = len(bits)
    edges = np.flatnonzero(np.diff(words)) + 1
    for a0, a1 in zip(np.concatenate([[0], edges]), np.concatenate([edges, [len(words)]])):
        sy = int(words[a0]); n = int(a1 - a0)
        while n > 0:
            k = min(n, 256); n -= k                  # one symbol followed by up to 255 repeats
            if use_rle and length[sy] + length[iM] + 8 < length[sy] * (k - 1):
                put(code[sy], length[sy]); put(code[iM], length[iM]); put(k - 1, 8)
            else:
                for _ in range(k):
                    put(code[sy], length[sy])
    data_bits = sum(n for _, n in bits[start:])
    v = np.array([b[0] for b in bits], np.uint64); n = np.array([b[1] for b in bits], np.int64)
    k = np.arange(58)[None, :]
    on = k < n[:, None]                                                  # bit k of entry i, most significant first
    shift = np.where(on, n[:, None] - 1 - k, 0).astype(np.uint64)
    stream = ((v[:, None] >> shift) & np.uint64(1)).astype(np.uint8)[on]
    payload = np.packbits(stream).tobytes()                              # zero-padded to a byte boundary
    return struct.pack("<IIIII", im, iM, (table_bits + 7) // 8, data_bits, 0) + payload


def piz_block(lines, words_per_pixel, use_rle=True):
    """lines: list (one per scanline) of lists (one per channel, file order) of uint16 word arrays (FLOAT / UINT pixels = 2 words,
    low word first) -> the block's compressed bytes"""
    planes = [np.stack([np.asarray(l[c], np.uint16) for l in lines]) for c in range(len(words_per_pixel))]   # (ny, nx * words per pixel)
    allw = np.concatenate([p.reshape(-1) for p in planes])
    used = np.zeros(65536, bool); used[allw] = True; used[0] = False                                # zero is always in the table, never in the bitmap
    bitmap = np.packbits(used.reshape(-1, 8)[:, ::-1], axis=1).reshape(-1)                          # bit (v & 7) of byte (v >> 3)
    nz = np.flatnonzero(bitmap)
    mn, mx = (int(nz[0]), int(nz[-1])) if len(nz) else (8191, 0)
    present = used.copy(); present[0] = True
    lut = np.zeros(65536, np.int64); lut[present] = np.arange(int(present.sum()))
    max_value = int(present.sum()) - 1
    coded = []
    for p, wpp in zip(planes, words_per_pixel):
        ny, n = p.shape
        t = lut[p].reshape(ny, n // wpp, wpp)
        for j in range(wpp):                        # the low and the high words of a wide channel are transformed as separate images
            t[:, :, j] = _piz_wavelet(t[:, :, j], max_value)
        coded.append(t.reshape(-1))
    huf = _piz_huffman(np.concatenate(coded), use_rle)
    return struct.pack("<HH", mn, mx) + (bitmap[mn:mx + 1].tobytes() if mn <= mx else b"") + struct.pack("<i", len(huf)) + huf


def exr_bytes(rgba, channels="RGB", pixel_type="float", compression="none", piz_rle=True, tiles=None, levels="one", line_order=0):
    """(H, W, 4) float32 -> OpenEXR bytes.  channels: subset of 'ABGR' letters; pixel_type float|half; compression none|zips|zip|piz.
    Scanline file by default; tiles=(tw, th) writes a single-part TILED file (version bit 0x200, `tiles` attribute, one chunk per tile with
    its {tile x, tile y, level x, level y} coordinates), levels = "one", "mipmap" or "ripmap" (box-filtered lower levels, rounded down, which a reader of the
    full-resolution image has to step over); line_order 1 = DECREASING_Y (chunks stored bottom-up).  Written from the OpenEXR file-layout
    document, independently of the C++ reader."""
    a = np.asarray(rgba, np.float32)
    h, w, _ = a.shape
    names = sorted(channels)
    ptype = {"float": 2, "half": 1}[pixel_type]
    comp = {"none": 0, "zips": 2, "zip": 3, "piz": 4}[compression]

    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(data)) + data
    chl = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", ptype, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2 | (0x200 if tiles else 0)) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp])) \
        + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", bytes([line_order])) \
        + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) \
        + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    if tiles:
        hdr += attr("tiles", "tiledesc", struct.pack("<IIB", tiles[0], tiles[1], {"one": 0, "mipmap": 1, "ripmap": 2}[levels]))   # rounding mode ROUND_DOWN (high nibble 0)
    hdr += b"\0"
    col = {"R": 0, "G": 1, "B": 2, "A": 3, "Y": 0}

    def pack(img, y0, y1, x0, x1):
        """the pixels [y0, y1) x [x0, x1) of img as a chunk's data: per line the channels in file order; compressed when that is smaller"""
        raw = bytearray()
        rows = []
        for y in range(y0, y1):
            rows.append([])
            for n in names:
                v = img[y, x0:x1, col[n]]
                with np.errstate(over="ignore"):    # (values beyond HALF's range become infinities, as in OpenEXR's float -> half conversion)
                    enc = (v.astype(np.float16) if ptype == 1 else v).tobytes()
                raw += enc
                rows[-1].append(np.frombuffer(enc, "<u2"))
        data = bytes(raw)
        if comp == 4:
            z = piz_block(rows, [1 if ptype == 1 else 2] * len(names), piz_rle)
            if len(z) < len(data):
                data = z
        elif comp:
            b = np.frombuffer(data, np.uint8)
            half = (len(b) + 1) // 2
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int16)
            assert len(b[0::2]) == half
            d = t.copy(); d[1:] = (t[1:] - t[:-1] + 128 + 256) % 256
            z = zlib.compress(d.astype(np.uint8).tobytes())
            if len(z) < len(data):
                data = z
        return data

    chunks = []          # (chunk header bytes, data) in offset-table order
    if tiles and levels == "ripmap":      # every (lx, ly) combination of halvings, ly outer, lx inner; level (0, 0) first
        tw, th = tiles

        def half(im, axis):
            n = im.shape[axis]; m = max(n // 2, 1)
            i0 = [min(2 * k, n - 1) for k in range(m)]; i1 = [min(2 * k + 1, n - 1) for k in range(m)]
            return (np.take(im, i0, axis) + np.take(im, i1, axis)) * 0.5
        row, ly = a, 0
        while True:
            img, lx = row, 0
            while True:
                lh, lw = img.shape[:2]
                for ty, y0 in enumerate(range(0, lh, th)):
                    for tx, x0 in enumerate(range(0, lw, tw)):
                        chunks.append((struct.pack("<iiii", tx, ty, lx, ly), pack(img, y0, min(y0 + th, lh), x0, min(x0 + tw, lw))))
                if lw == 1:
                    break
                img = half(img, 1); lx += 1
            if row.shape[0] == 1:
                break
            row = half(row, 0); ly += 1
    elif tiles:
        tw, th = tiles
        img, lvl = a, 0
        while True:
            lh, lw = img.shape[:2]
            ys = list(range(0, lh, th))
            for ty, y0 in enumerate(ys):
                for tx, x0 in enumerate(range(0, lw, tw)):
                    chunks.append((struct.pack("<iiii", tx, ty, lvl, lvl), pack(img, y0, min(y0 + th, lh), x0, min(x0 + tw, lw))))
            if levels == "one" or (lw == 1 and lh == 1):
                break
            nh, nw = max(lh // 2, 1), max(lw // 2, 1)                      # ROUND_DOWN
            img = np.stack([img[min(2 * y, lh - 1)] for y in range(nh)])[:, [min(2 * x, lw - 1) for x in range(nw)]] * 0.5 \
                + np.stack([img[min(2 * y + 1, lh - 1)] for y in range(nh)])[:, [min(2 * x + 1, lw - 1) for x in range(nw)]] * 0.5
            lvl += 1
    else:
        lines = {0: 1, 2: 1, 3: 16, 4: 32}[comp]
        for y0 in range(0, h, lines):
            chunks.append((struct.pack("<i", y0), pack(a, y0, min(y0 + lines, h), 0, w)))
    # the offset table is always in increasing-y (and level) order; DECREASING_Y only changes where the chunks lie in the file
    order = list(range(len(chunks)))
    stored = order[::-1] if line_order == 1 else order
    off = len(hdr) + 8 * len(chunks)
    offs, body = [0] * len(chunks), bytearray()
    for k in stored:
        head, data = chunks[k]
        offs[k] = off + len(body)
        body += head + struct.pack("<i", len(data)) + data
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
    col = {"R": 0, "G": 1, "B": 2, "A": 3, "Y": 0}
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
