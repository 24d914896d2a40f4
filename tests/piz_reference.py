"""A second PIZ DECODER, in plain Python: test infrastructure that cross-checks the numpy PIZ encoder of tests/assets.py with something other than the product's
reader (moonshine_amd/host/exr.cpp), so a misreading shared by that encoder and that reader cannot pass unnoticed.

Written from the published algorithm of OpenEXR's PIZ compression as its reference implementation states it (ImfPizCompressor: bitmap -> reverse lookup table,
ImfHuf hufUncompress: packed 6-bit code lengths with zero-run codes 59..63, canonical codes assigned from the longest length down, most-significant-bit-first
bit stream, run-length symbol = iM followed by an 8-bit count; ImfWav wav2Decode: the inverse of the 2-D two-tap wavelet from the coarsest level to the finest,
wdec14 for data below 2^14 values, wdec16 otherwise) — decoding direction only, pointer-walking loops, no code shared with either of the other two.
Scanline files only (blocks of 32 lines), HALF and FLOAT channels."""
import struct

import numpy as np


class _Bits:
    """most-significant-bit-first reader over bytes"""

    def __init__(self, data, pos=0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n):
        while self.lc < n:
            self.c = ((self.c << 8) | self.d[self.p]) & ((1 << 128) - 1); self.p += 1; self.lc += 8
        self.lc -= n
        return (self.c >> self.lc) & ((1 << n) - 1)


def _huf_uncompress(data, n_raw):
    im, iM, table_len, n_bits, _ = struct.unpack_from("<IIIII", data, 0)
    assert im <= 65536 and iM <= 65536
    # code lengths im..iM, 6 bits each; 63 + 8 bits = a run of 6..261 zeros, 59..62 = 2..5 zeros
    length = [0] * 65537
    br = _Bits(data, 20)
    s = im
    while s <= iM:
        l = br.get(6)
        if l == 63:
            s += br.get(8) + 6
        elif l >= 59:
            s += l - 59 + 2
        else:
            length[s] = l; s += 1
    start = br.p                                    # the coded words begin at the next byte boundary
    assert table_len == start - 20, "the table's byte count in the header is the bytes the lengths take"
    # canonical codes: per length, consecutive values; the first value of a length from the counts of the LONGER lengths
    count = [0] * 59
    for l in length:
        count[l] += 1
    first = [0] * 59
    c = 0
    for l in range(58, 0, -1):
        first[l] = c; c = (c + count[l]) >> 1
    table = {}
    for sym in range(65537):
        l = length[sym]
        if l:
            table[(l, first[l])] = sym; first[l] += 1
    out = []
    br = _Bits(data, start)
    left = n_bits
    while left > 0 and len(out) < n_raw:
        code, l = 0, 0
        while True:
            code = (code << 1) | br.get(1); l += 1; left -= 1
            if (l, code) in table:
                break
            assert l < 59 and left >= 0, "no code matches"
        sym = table[(l, code)]
        if sym == iM:                               # run-length symbol: repeat the previous word
            n = br.get(8); left -= 8
            assert out, "run without a word before it"
            out += [out[-1]] * n
        else:
            out.append(sym)
    assert len(out) == n_raw, (len(out), n_raw)
    return out


def _wdec14(l, h):
    ls = l - 0x10000 if l >= 0x8000 else l
    hs = h - 0x10000 if h >= 0x8000 else h
    ai = ls + (hs & 1) + (hs >> 1)
    return ai & 0xffff, (ai - hs) & 0xffff


def _wdec16(l, h):
    bb = (l - (h >> 1)) & 0xffff
    aa = (h + bb - 0x8000) & 0xffff
    return aa, bb


def _wav2_decode(buf, base, nx, ox, ny, oy, max_value):
    dec = _wdec14 if max_value < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py, ey = base, base + oy * (ny - p2)
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01, p10 = px + ox1, px + oy1
                p11 = p10 + ox1
                i00, i10 = dec(buf[px], buf[p10])
                i01, i11 = dec(buf[p01], buf[p11])
                buf[px], buf[p01] = dec(i00, i01)
                buf[p10], buf[p11] = dec(i10, i11)
                px += ox2
            if nx & p:
                p10 = px + oy1
                buf[px], buf[p10] = dec(buf[px], buf[p10])
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                p01 = px + ox1
                buf[px], buf[p01] = dec(buf[px], buf[p01])
                px += ox2
        p2 = p
        p >>= 1


def _piz_block(data, nx, ny, sizes):
    """-> list of uint16 words in scanline order (per line: every channel's nx * size words)"""
    mn, mx = struct.unpack_from("<HH", data, 0)
    bitmap = bytearray(8192)
    pos = 4
    if mn <= mx:
        bitmap[mn:mx + 1] = data[pos:pos + mx - mn + 1]; pos += mx - mn + 1
    lut = [i for i in range(65536) if i == 0 or (bitmap[i >> 3] >> (i & 7)) & 1]
    max_value = len(lut) - 1
    (length,) = struct.unpack_from("<i", data, pos); pos += 4
    total = sum(nx * ny * s for s in sizes)
    buf = _huf_uncompress(data[pos:pos + length], total)
    start, planes = 0, []
    for s in sizes:                                  # the buffer holds the channels one after the other; a wide channel's low and high words alternate
        for j in range(s):
            _wav2_decode(buf, start + j, nx, s, ny, nx * s, max_value)
        planes.append(start); start += nx * ny * s
    buf = [lut[v] for v in buf]
    out = []
    for y in range(ny):
        for s, st in zip(sizes, planes):
            out += buf[st + y * nx * s: st + (y + 1) * nx * s]
    return out


def exr_piz_decode(data):
    """scanline PIZ OpenEXR bytes -> {channel name: (H, W) array} of float16 / float32 as stored"""
    assert struct.unpack_from("<I", data, 0)[0] == 20000630
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
    assert attrs["compression"][0] == 4, "not a PIZ file"
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    sizes = [1 if t == 1 else 2 for _, t in chans]
    nblocks = (h + 31) // 32
    offs = struct.unpack_from("<%dQ" % nblocks, data, pos)
    out = {nm: np.zeros((h, w), np.float16 if t == 1 else np.float32) for nm, t in chans}
    for o in offs:
        by, n = struct.unpack_from("<ii", data, o)
        ny = min(32, y1 + 1 - by)
        raw_len = sum(sizes) * 2 * w * ny
        blk = data[o + 8:o + 8 + n]
        words = np.frombuffer(blk, "<u2").tolist() if n == raw_len else _piz_block(blk, w, ny, sizes)
        words = np.asarray(words, np.uint16)
        k = 0
        for y in range(ny):
            for (nm, t), s in zip(chans, sizes):
                row = words[k:k + w * s]; k += w * s
                out[nm][by - y0 + y] = row.view(np.float16) if t == 1 else row.astype("<u2").view("<f4")
    return out
