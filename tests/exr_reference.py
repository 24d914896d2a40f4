"""A second OpenEXR READER, in plain Python: test infrastructure.  The product's reader (moonshine_amd/host/exr.cpp) is pinned by the numpy writer of tests/assets.py;
this module reads the same files from the other side — scanline and single-part TILED files (ONE_LEVEL / MIPMAP / RIPMAP, ROUND_DOWN), NONE / ZIPS / ZIP / PIZ, HALF /
FLOAT — written from the OpenEXR file-layout document, and unlike a reader that simply follows each chunk's own coordinates it REQUIRES the offset table to list
the chunks in the order the document prescribes (scanline blocks by increasing y; tiles level by level, row-major inside a level; rip-map levels with ly outer and
lx inner) and every tile to have exactly the clipped size.  PIZ blocks go through tests/piz_reference.py.  Only level (0, 0) is returned."""
import struct
import zlib

import numpy as np

import piz_reference


def _undo_zip(raw, expected):
    b = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int64)
    assert len(b) == expected
    d = np.cumsum(b - 128) + 128                      # d[i] = d[i-1] + b[i] - 128, d[0] = b[0]
    d = (d & 0xff).astype(np.uint8)
    half = (expected + 1) // 2
    out = np.zeros(expected, np.uint8)
    out[0::2] = d[:half]; out[1::2] = d[half:]
    return out.tobytes()


def read(data):
    """-> {channel: (H, W) array as stored (float16 / float32)}"""
    magic, version = struct.unpack_from("<II", data, 0)
    assert magic == 20000630 and (version & 0xff) == 2
    tiled = bool(version & 0x200)
    pos, attrs = 8, {}
    while data[pos] != 0:
        e = data.index(b"\0", pos); name = data[pos:e].decode(); pos = e + 1
        e = data.index(b"\0", pos); pos = e + 1
        n = struct.unpack_from("<I", data, pos)[0]; pos += 4
        attrs[name] = data[pos:pos + n]; pos += n
    pos += 1
    chans, c, p = [], attrs["channels"], 0
    while c[p] != 0:
        e = c.index(b"\0", p); nm = c[p:e].decode(); p = e + 1
        chans.append((nm, struct.unpack_from("<i", c, p)[0])); p += 16
    assert [n for n, _ in chans] == sorted(n for n, _ in chans), "channels are stored in alphabetical order"
    comp = attrs["compression"][0]
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    sizes = [1 if t == 1 else 2 for _, t in chans]
    bpp = sum(sizes) * 2
    out = {nm: np.zeros((h, w), np.float16 if t == 1 else np.float32) for nm, t in chans}

    def put(block, bx, by, bw, bh):
        raw_len = bpp * bw * bh
        if len(block) == raw_len:
            raw = block
        elif comp in (2, 3):
            raw = _undo_zip(block, raw_len)
        elif comp == 4:
            raw = np.asarray(piz_reference._piz_block(block, bw, bh, sizes), "<u2").tobytes()
        else:
            raise AssertionError("compression %d" % comp)
        words = np.frombuffer(raw, "<u2")
        k = 0
        for y in range(bh):
            for (nm, t), s in zip(chans, sizes):
                row = words[k:k + bw * s]; k += bw * s
                out[nm][by + y, bx:bx + bw] = row.view(np.float16) if t == 1 else row.view("<f4")

    if not tiled:
        lines = {0: 1, 2: 1, 3: 16, 4: 32}[comp]
        nblocks = (h + lines - 1) // lines
        offs = struct.unpack_from("<%dQ" % nblocks, data, pos)
        for i, o in enumerate(offs):
            by, n = struct.unpack_from("<ii", data, o)
            assert by == y0 + i * lines, "scanline blocks are listed by increasing y"
            put(data[o + 8:o + 8 + n], 0, by - y0, w, min(lines, h - (by - y0)))
        return out
    tw, th, mode = struct.unpack("<IIB", attrs["tiles"])
    assert mode >> 4 == 0, "ROUND_DOWN only"
    mode &= 0xf

    def levels(n):                                     # ROUND_DOWN: sizes n, n // 2, ... down to 1
        v = [n]
        while v[-1] > 1:
            v.append(max(v[-1] // 2, 1))
        return v
    if mode == 0:
        order = [(0, 0, w, h)]
    elif mode == 1:
        lw, lh = levels(w), levels(h)
        nl = max(len(lw), len(lh))
        order = [(l, l, lw[min(l, len(lw) - 1)], lh[min(l, len(lh) - 1)]) for l in range(nl)]
    else:
        order = [(lx, ly, ww, hh) for ly, hh in enumerate(levels(h)) for lx, ww in enumerate(levels(w))]
    expected = [(tx, ty, lx, ly, ww, hh) for lx, ly, ww, hh in order for ty in range((hh + th - 1) // th) for tx in range((ww + tw - 1) // tw)]
    offs = struct.unpack_from("<%dQ" % len(expected), data, pos)
    assert pos + 8 * len(expected) <= min(offs), "the offset table has one entry per tile of every level"
    for o, (tx, ty, lx, ly, ww, hh) in zip(offs, expected):
        assert struct.unpack_from("<iiii", data, o) == (tx, ty, lx, ly), "tile order of the offset table"
        (n,) = struct.unpack_from("<i", data, o + 16)
        if (lx, ly) == (0, 0):
            bw, bh = min(tw, ww - tx * tw), min(th, hh - ty * th)
            put(data[o + 20:o + 20 + n], tx * tw, ty * th, bw, bh)
    return out
