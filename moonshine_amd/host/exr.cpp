// exr.cpp — OpenEXR scanline reader/writer for the hot path's I/O contract (engine/fileformats/exr.zig:126-231,
// which wraps syoyo/tinyexr — un-vendored in the reference, deps/tinyexr is empty; this is a from-scratch codec of
// the published OpenEXR 2 file layout).
//   load: RGBA f32, like tinyexr's LoadEXRFromMemory (exr.zig:208-229): channels R,G,B,(A) by name, HALF or FLOAT
//         or UINT pixels, compression NONE / RLE / ZIPS / ZIP / PIZ (tinyexr's set), single-part scanline files, any line order.
//   save: three FLOAT channels in header order B,G,R, scanline, ZIP (exr.zig:137-206 with tinyexr's header defaults; alpha is dropped).
#include "host.h"
#include <zlib.h>
#include <atomic>
#include <thread>
#include <system_error>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>

namespace msne_host {

static float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 0x1f, m = h & 0x3ff;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else { float f = (float)m * 0x1p-24f; if (h >> 15) f = -f; memcpy(&u, &f, 4); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

struct Reader {
    const uint8_t* p; size_t n, pos = 0; bool ok = true;
    bool need(size_t k) { if (pos > n || k > n - pos) { ok = false; return false; } return true; }   // (pos + k may wrap: offsets come from the file)
    uint8_t u8() { if (!need(1)) return 0; return p[pos++]; }
    uint32_t u32() { if (!need(4)) return 0; uint32_t v; memcpy(&v, p + pos, 4); pos += 4; return v; }
    int32_t i32() { return (int32_t)u32(); }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v; memcpy(&v, p + pos, 8); pos += 8; return v; }
    std::string str() { std::string s; while (need(1) && p[pos]) s.push_back((char)p[pos++]); pos++; return s; }
};

struct Channel { std::string name; int type; int xs, ys; };

static bool inflate_all(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, size_t expect) {
    dst.resize(expect);
    uLongf len = (uLongf)expect;
    return uncompress(dst.data(), &len, src, (uLong)n) == Z_OK && len == expect;
}
static void undo_predictor_and_interleave(std::vector<uint8_t>& buf, std::vector<uint8_t>& out) {
    for (size_t i = 1; i < buf.size(); i++) buf[i] = (uint8_t)(buf[i - 1] + buf[i] - 128);
    out.resize(buf.size());
    const size_t half = (buf.size() + 1) / 2;
    for (size_t i = 0, a = 0, b = half; i < buf.size();) {
        out[i++] = buf[a++];
        if (i < buf.size()) out[i++] = buf[b++];
    }
}
static bool rle_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& dst, size_t expect) {
    dst.clear(); dst.reserve(expect);
    size_t i = 0;
    while (i < n) {
        const int8_t c = (int8_t)src[i++];
        if (c < 0) { const size_t k = (size_t)(-c); if (i + k > n) return false; dst.insert(dst.end(), src + i, src + i + k); i += k; }
        else { if (i >= n) return false; dst.insert(dst.end(), (size_t)c + 1, src[i]); i++; }
    }
    return dst.size() == expect;
}

// ---------------- PIZ (OpenEXR ImfPizCompressor / ImfHuf / ImfWav; a from-scratch decoder of the published scheme) ----------------
// A PIZ block holds up to 32 scanlines.  The encoder regroups the block channel by channel as 16-bit words (HALF = 1 word per
// pixel, FLOAT / UINT = 2), maps the words that occur onto 0..k through a bitmap-derived table, applies a 2-D Haar-style
// wavelet per channel (and per 16-bit half of wide types), and Huffman-codes all words with canonical codes + a run-length
// escape.  Decoding runs those four steps backwards.
namespace piz {

constexpr int ENC_SIZE = (1 << 16) + 1;         // symbols 0..65535 + the run-length escape
constexpr int SHORT_ZEROCODE_RUN = 59, LONG_ZEROCODE_RUN = 63, SHORTEST_LONG_RUN = 2 + LONG_ZEROCODE_RUN - SHORT_ZEROCODE_RUN;

struct BitReader {
    const uint8_t* p; size_t n; size_t pos = 0;   // pos in bits, MSB first
    bool get(int nbits, uint64_t& out) {
        if (nbits < 0 || nbits > 57 || pos + (size_t)nbits > n * 8) return false;
        uint64_t v = 0;
        for (int i = 0; i < nbits; i++, pos++) v = (v << 1) | ((p[pos >> 3] >> (7 - (pos & 7))) & 1u);
        out = v; return true;
    }
};

// code lengths of symbols im..iM: 6 bits each, 59..62 = a run of 2..5 zero lengths, 63 + 8 bits = a run of 6..261
static bool unpack_lengths(BitReader& br, uint32_t im, uint32_t iM, std::vector<uint8_t>& len) {
    len.assign(ENC_SIZE, 0);
    for (uint32_t s = im; s <= iM; s++) {
        uint64_t l;
        if (!br.get(6, l)) return false;
        if (l == (uint64_t)LONG_ZEROCODE_RUN) {
            uint64_t r; if (!br.get(8, r)) return false;
            const uint64_t run = r + SHORTEST_LONG_RUN;
            if (s + run > (uint64_t)iM + 1) return false;
            s += (uint32_t)run - 1;
        } else if (l >= (uint64_t)SHORT_ZEROCODE_RUN) {
            const uint64_t run = l - SHORT_ZEROCODE_RUN + 2;
            if (s + run > (uint64_t)iM + 1) return false;
            s += (uint32_t)run - 1;
        } else len[s] = (uint8_t)l;
    }
    return true;
}

// canonical codes: per length the first code value (longer codes take the numerically smaller values) and the symbols in index order
struct Canon { uint64_t first[59]; uint32_t count[59]; uint32_t offset[59]; std::vector<uint32_t> syms; int min_len = 59, max_len = 0; };
static bool build_canon(const std::vector<uint8_t>& len, Canon& c) {
    uint64_t n[59] = { 0 };
    for (int s = 0; s < ENC_SIZE; s++) { if (len[s] > 58) return false; n[len[s]]++; }
    for (int l = 0; l < 59; l++) c.count[l] = (uint32_t)n[l];
    uint64_t code = 0;
    for (int l = 58; l >= 1; l--) { const uint64_t next = (code + n[l]) >> 1; c.first[l] = code; code = next; }
    uint32_t off = 0;
    for (int l = 1; l <= 58; l++) { c.offset[l] = off; off += c.count[l]; if (c.count[l]) { c.min_len = std::min(c.min_len, l); c.max_len = std::max(c.max_len, l); } }
    c.syms.resize(off);
    uint32_t fill[59] = { 0 };
    for (int s = 0; s < ENC_SIZE; s++) if (len[s]) c.syms[c.offset[len[s]] + fill[len[s]]++] = (uint32_t)s;
    return true;
}

static bool huf_uncompress(const uint8_t* src, size_t n, std::vector<uint16_t>& out, size_t n_raw) {
    out.assign(n_raw, 0);
    if (n == 0) return n_raw == 0;
    if (n < 20) return false;
    uint32_t im, iM, nbits; memcpy(&im, src, 4); memcpy(&iM, src + 4, 4); memcpy(&nbits, src + 12, 4);
    if (im >= (uint32_t)ENC_SIZE || iM >= (uint32_t)ENC_SIZE || im > iM) return false;
    BitReader tb{ src + 20, n - 20 };
    std::vector<uint8_t> len;
    if (!unpack_lengths(tb, im, iM, len)) return false;
    const size_t table_bytes = (tb.pos + 7) / 8;
    if (20 + table_bytes > n || (size_t)nbits > 8 * (n - 20 - table_bytes)) return false;
    Canon c;
    if (!build_canon(len, c) || c.max_len == 0) return false;
    // fast table for codes of up to 12 bits; longer ones continue bit by bit
    constexpr int FAST = 12;
    std::vector<uint32_t> fast((size_t)1 << FAST, 0);          // (symbol << 6) | length, 0 = not a short code
    for (int l = c.min_len; l <= std::min(FAST, c.max_len); l++)
        for (uint32_t k = 0; k < c.count[l]; k++) {
            const uint64_t code = c.first[l] + k;
            if (code >> l) return false;
            const uint32_t e = (c.syms[c.offset[l] + k] << 6) | (uint32_t)l;
            for (uint64_t f = code << (FAST - l), end = (code + 1) << (FAST - l); f < end; f++) fast[(size_t)f] = e;
        }
    BitReader br{ src + 20 + table_bytes, (size_t)(nbits + 7) / 8 };
    const size_t total = nbits;
    size_t o = 0;
    auto bit_at = [&](size_t pos) -> uint32_t { return pos < total ? (br.p[pos >> 3] >> (7 - (pos & 7))) & 1u : 0u; };
    while (br.pos < total) {
        uint32_t sym = 0; int l = 0;
        // the next FAST bits (zero-padded past the end of the stream): one unaligned big-endian window instead of a bit at a time
        uint32_t peek;
        {
            const size_t byte = br.pos >> 3;
            uint32_t w = 0;
            for (size_t k = 0; k < 4; k++) w = (w << 8) | (byte + k < br.n ? br.p[byte + k] : 0u);
            peek = (w << (br.pos & 7)) >> (32 - FAST);
            if (br.pos + FAST > total) peek &= ~0u << (br.pos + FAST - total);   // bits past `total` inside the last byte read as 0, like bit_at()
        }
        const uint32_t e = fast[peek];
        if (e && br.pos + (e & 63u) <= total) { sym = e >> 6; l = (int)(e & 63u); }
        else {
            uint64_t v = 0; bool found = false;
            for (l = 1; l <= c.max_len && br.pos + (size_t)l <= total; l++) {
                v = (v << 1) | bit_at(br.pos + (size_t)l - 1);
                if (c.count[l] && v >= c.first[l] && v - c.first[l] < c.count[l]) { sym = c.syms[c.offset[l] + (uint32_t)(v - c.first[l])]; found = true; break; }
            }
            if (!found) return false;
        }
        br.pos += (size_t)l;
        if (sym == iM) {                      // run-length escape: repeat the previous word
            uint64_t run;
            if (br.pos + 8 > total || !br.get(8, run) || o == 0 || o + run > n_raw) return false;
            const uint16_t prev = out[o - 1];
            for (uint64_t k = 0; k < run; k++) out[o++] = prev;
        } else {
            if (o >= n_raw || sym > 0xffffu) return false;
            out[o++] = (uint16_t)sym;
        }
    }
    return o == n_raw;
}

// inverse wavelet.  14-bit variant (all values < 2^14): signed arithmetic; 16-bit variant: modulo arithmetic
static inline void wdec14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t)(int16_t)ai; b = (uint16_t)(int16_t)(ai - hs);
}
static inline void wdec16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    a = (uint16_t)((d + bb - 0x8000) & 0xffff); b = (uint16_t)bb;
}
static void wav2_decode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    int p2 = p; p >>= 1;
    while (p >= 1) {
        uint16_t* py = in; uint16_t* const ey = in + (ptrdiff_t)oy * (ny - p2);
        const ptrdiff_t oy1 = (ptrdiff_t)oy * p, oy2 = (ptrdiff_t)oy * p2, ox1 = (ptrdiff_t)ox * p, ox2 = (ptrdiff_t)ox * p2;
        uint16_t i00, i01, i10, i11;
        for (; py <= ey; py += oy2) {
            uint16_t* px = py; uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) {
                uint16_t* p01 = px + ox1; uint16_t* p10 = px + oy1; uint16_t* p11 = p10 + ox1;
                if (w14) { wdec14(*px, *p10, i00, i10); wdec14(*p01, *p11, i01, i11); wdec14(i00, i01, *px, *p01); wdec14(i10, i11, *p10, *p11); }
                else { wdec16(*px, *p10, i00, i10); wdec16(*p01, *p11, i01, i11); wdec16(i00, i01, *px, *p01); wdec16(i10, i11, *p10, *p11); }
            }
            if (nx & p) { uint16_t* p10 = px + oy1; if (w14) wdec14(*px, *p10, i00, *p10); else wdec16(*px, *p10, i00, *p10); *px = i00; }
        }
        if (ny & p) {
            uint16_t* px = py; uint16_t* const ex = py + (ptrdiff_t)ox * (nx - p2);
            for (; px <= ex; px += ox2) { uint16_t* p01 = px + ox1; if (w14) wdec14(*px, *p01, i00, *p01); else wdec16(*px, *p01, i00, *p01); *px = i00; }
        }
        p2 = p; p >>= 1;
    }
}

// one block -> raw scanline layout (per line: the channels in file order, each W pixels)
static bool decode_block(const uint8_t* src, size_t n, const std::vector<int>& words_per_pixel, size_t W, size_t nlines, std::vector<uint8_t>& raw) {
    size_t total = 0;
    for (int wpp : words_per_pixel) total += (size_t)wpp * W * nlines;
    if (n < 4) return false;
    uint16_t min_nz, max_nz; memcpy(&min_nz, src, 2); memcpy(&max_nz, src + 2, 2);
    size_t pos = 4;
    std::vector<uint8_t> bitmap(8192, 0);
    if (max_nz >= 8192) return false;
    if (min_nz <= max_nz) {
        const size_t k = (size_t)max_nz - min_nz + 1;
        if (pos + k > n) return false;
        memcpy(&bitmap[min_nz], src + pos, k); pos += k;
    }
    std::vector<uint16_t> lut(65536, 0);
    uint32_t k = 0;
    for (uint32_t i = 0; i < 65536; i++) if (i == 0 || (bitmap[i >> 3] & (1u << (i & 7)))) lut[k++] = (uint16_t)i;
    const uint16_t max_value = (uint16_t)(k - 1);
    if (pos + 4 > n) return false;
    int32_t hlen; memcpy(&hlen, src + pos, 4); pos += 4;
    if (hlen < 0 || (size_t)hlen > n - pos) return false;
    std::vector<uint16_t> buf;
    if (!huf_uncompress(src + pos, (size_t)hlen, buf, total)) return false;
    size_t at = 0;
    std::vector<size_t> start(words_per_pixel.size());
    for (size_t c = 0; c < words_per_pixel.size(); c++) {
        start[c] = at;
        for (int j = 0; j < words_per_pixel[c]; j++) wav2_decode(buf.data() + at + j, (int)W, words_per_pixel[c], (int)nlines, (int)(W * (size_t)words_per_pixel[c]), max_value);
        at += (size_t)words_per_pixel[c] * W * nlines;
    }
    for (auto& v : buf) v = lut[v];
    raw.resize(total * 2);
    size_t o = 0;
    for (size_t y = 0; y < nlines; y++)
        for (size_t c = 0; c < words_per_pixel.size(); c++) {
            const size_t cnt = (size_t)words_per_pixel[c] * W;
            memcpy(&raw[o], buf.data() + start[c] + y * cnt, cnt * 2); o += cnt * 2;
        }
    return true;
}

}  // namespace piz

bool exr_load(const std::string& path, Image& img, std::string& err) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) { err = "cannot read " + path; return false; }
    Reader r{ file.data(), file.size() };
    if (r.u32() != 20000630u) { err = "not an OpenEXR file"; return false; }
    const uint32_t version = r.u32();
    if ((version & 0xff) != 2) { err = "unsupported EXR version"; return false; }
    const bool tiled = (version & 0x200u) != 0;   // single-part tiled file: tinyexr's LoadEXRFromMemory (exr.zig:109-110) reads those too
    if (version & 0x1800u) { err = "multi-part / deep EXR files are not supported"; return false; }
    std::vector<Channel> channels; int compression = 0; int32_t dw[4] = { 0, 0, -1, -1 };
    uint32_t tile_w = 0, tile_h = 0; int level_mode = 0; bool have_tiles = false;
    for (;;) {
        const std::string name = r.str();
        if (!r.ok) { err = "truncated EXR header"; return false; }
        if (name.empty()) break;
        const std::string type = r.str();
        const uint32_t size = r.u32();
        if (!r.need(size)) { err = "truncated EXR header"; return false; }
        Reader a{ r.p + r.pos, size };
        if (name == "channels") {
            for (;;) {
                Channel c; c.name = a.str();
                if (c.name.empty() || !a.ok) break;
                c.type = a.i32(); a.u8(); a.u8(); a.u8(); a.u8(); c.xs = a.i32(); c.ys = a.i32();
                channels.push_back(c);
            }
        } else if (name == "compression") compression = a.u8();
        else if (name == "dataWindow") { for (int k = 0; k < 4; k++) dw[k] = a.i32(); }
        else if (name == "tiles") { tile_w = a.u32(); tile_h = a.u32(); level_mode = a.u8() & 15; have_tiles = a.ok; }
        r.pos += size;
    }
    const int64_t W = (int64_t)dw[2] - dw[0] + 1, H = (int64_t)dw[3] - dw[1] + 1;
    if (W <= 0 || H <= 0 || W > 65536 || H > 65536 || channels.empty()) { err = "bad EXR header"; return false; }
    for (auto& c : channels) if (c.xs != 1 || c.ys != 1) { err = "subsampled EXR channels are not supported"; return false; }
    if (tiled && (!have_tiles || tile_w == 0 || tile_h == 0 || tile_w > 65536 || tile_h > 65536 || level_mode > 2)) { err = "bad EXR tile description"; return false; }
    int lines_per_block;
    switch (compression) {
        case 0: case 1: case 2: lines_per_block = 1; break;
        case 3: lines_per_block = 16; break;
        case 4: lines_per_block = 32; break;
        default: err = "EXR compression " + std::to_string(compression) + " is not supported (NONE, RLE, ZIPS, ZIP, PIZ are)"; return false;
    }
    // chunks of the full-resolution image: scanline blocks, or the tiles of level (0, 0) — in the offset table the first tiles_x * tiles_y entries
    // for every level mode (ONE_LEVEL, MIPMAP_LEVELS: level 0 first; RIPMAP_LEVELS: (0, 0) first); lower-resolution levels are not read, as
    // tinyexr's simple loader keeps only the first image
    const size_t tiles_x = tiled ? (size_t)((W + tile_w - 1) / tile_w) : 1, tiles_y = tiled ? (size_t)((H + tile_h - 1) / tile_h) : 0;
    const size_t nblocks = tiled ? tiles_x * tiles_y : (size_t)((H + lines_per_block - 1) / lines_per_block);
    if (r.pos > file.size() || nblocks > (file.size() - r.pos) / 8) { err = "truncated EXR offset table"; return false; }   // before anything is sized by the header
    std::vector<uint64_t> offsets(nblocks);
    for (auto& o : offsets) o = r.u64();
    if (!r.ok) { err = "truncated EXR offset table"; return false; }
    size_t px_bytes = 0;
    for (size_t c = 0; c < channels.size(); c++) px_bytes += channels[c].type == 1 ? 2 : 4;
    // which channel feeds which of R,G,B,A.  tinyexr's LoadEXR: a file with ONE channel, whatever its name, is a grey image whose value goes to all four components;
    // otherwise by name, A optional (1.0).  (More lenient than tinyexr, which refuses a file without R, G or B: a missing colour channel reads as 0, a luminance
    // channel "Y" next to others as grey.)
    int src[4] = { -1, -1, -1, -1 };
    for (size_t c = 0; c < channels.size(); c++) {
        const std::string& nm = channels[c].name;
        if (nm == "R") src[0] = (int)c; else if (nm == "G") src[1] = (int)c; else if (nm == "B") src[2] = (int)c; else if (nm == "A") src[3] = (int)c;
    }
    if (channels.size() == 1) src[0] = src[1] = src[2] = src[3] = 0;
    else if (src[0] < 0 && src[1] < 0 && src[2] < 0) { if (channels[0].name == "Y") src[0] = src[1] = src[2] = 0; else { err = "EXR has no R/G/B channels"; return false; } }
    // a header may claim more pixels than the file can hold: deflate expands at most ~1032:1, RLE 64:1
    if ((double)px_bytes * (double)W * (double)H > (double)file.size() * 1100.0) { err = "EXR data window exceeds the file's data"; return false; }
    img.w = (uint32_t)W; img.h = (uint32_t)H; img.rgba.assign((size_t)W * H * 4, 0.0f);
    for (size_t i = 0; i < (size_t)W * H; i++) img.rgba[4 * i + 3] = 1.0f;
    std::vector<uint8_t> tmp, raw;
    for (size_t b = 0; b < nblocks; b++) {
        if (offsets[b] >= file.size()) { err = "EXR chunk offset outside the file"; return false; }
        Reader c{ file.data(), file.size(), (size_t)offsets[b] };
        // the rectangle of the data window this chunk covers: [col0, col0 + ncols) x [row0, row0 + nlines)
        int64_t row0, col0 = 0; size_t nlines, ncols = (size_t)W;
        if (tiled) {
            const int32_t tx = c.i32(), ty = c.i32(), lx = c.i32(), ly = c.i32();
            if (!c.ok) { err = "truncated EXR chunk"; return false; }
            if (lx != 0 || ly != 0) { err = "EXR tile of a lower-resolution level where a full-resolution tile is expected"; return false; }
            if (tx < 0 || ty < 0 || (size_t)tx >= tiles_x || (size_t)ty >= tiles_y) { err = "EXR tile outside the data window"; return false; }
            col0 = (int64_t)tx * tile_w; row0 = (int64_t)ty * tile_h;
            ncols = (size_t)std::min<int64_t>(tile_w, W - col0); nlines = (size_t)std::min<int64_t>(tile_h, H - row0);
        } else {
            const int32_t y0 = c.i32();
            row0 = (int64_t)y0 - dw[1];
            if (!c.ok || row0 < 0 || row0 >= H) { err = "EXR chunk outside the data window"; return false; }
            nlines = (size_t)std::min<int64_t>(lines_per_block, H - row0);
        }
        const int32_t dsize = c.i32();
        if (!c.ok || dsize < 0 || !c.need((size_t)dsize)) { err = "truncated EXR chunk"; return false; }
        const size_t line_bytes = ncols * px_bytes;
        const size_t expect = nlines * line_bytes;
        const uint8_t* data = c.p + c.pos;
        if ((size_t)dsize == expect) { raw.assign(data, data + expect); }      // stored uncompressed (also when compression did not help)
        else if (compression == 1) { if (!rle_decode(data, (size_t)dsize, tmp, expect)) { err = "bad RLE data"; return false; } undo_predictor_and_interleave(tmp, raw); }
        else if (compression == 2 || compression == 3) { if (!inflate_all(data, (size_t)dsize, tmp, expect)) { err = "bad ZIP data"; return false; } undo_predictor_and_interleave(tmp, raw); }
        else if (compression == 4) {
            std::vector<int> wpp; for (auto& ch : channels) wpp.push_back(ch.type == 1 ? 1 : 2);
            if (!piz::decode_block(data, (size_t)dsize, wpp, ncols, nlines, raw) || raw.size() != expect) { err = "bad PIZ data"; return false; }
        }
        else { err = "EXR chunk size mismatch"; return false; }
        for (size_t l = 0; l < nlines; l++) {
            const uint8_t* line = raw.data() + l * line_bytes;
            float* out = &img.rgba[((size_t)(row0 + (int64_t)l) * W + (size_t)col0) * 4];
            size_t ch_at = 0;
            for (size_t ci = 0; ci < channels.size(); ci++) {
                const Channel& ch = channels[ci];
                const uint8_t* q = line + ch_at;
                ch_at += ncols * (ch.type == 1 ? 2 : 4);
                for (int k = 0; k < 4; k++) {
                    if (src[k] != (int)ci) continue;
                    for (size_t x = 0; x < ncols; x++) {
                        float v;
                        if (ch.type == 1) { uint16_t hv; memcpy(&hv, q + 2 * x, 2); v = half_to_float(hv); }
                        else if (ch.type == 2) memcpy(&v, q + 4 * x, 4);
                        else { uint32_t uv; memcpy(&uv, q + 4 * x, 4); v = (float)uv; }
                        out[4 * x + k] = v;
                    }
                }
            }
        }
    }
    return true;
}

static void put_str(std::vector<uint8_t>& o, const char* s) { o.insert(o.end(), s, s + strlen(s) + 1); }
template <typename T> static void put(std::vector<uint8_t>& o, T v) { const uint8_t* p = (const uint8_t*)&v; o.insert(o.end(), p, p + sizeof(T)); }
static void put_attr(std::vector<uint8_t>& o, const char* name, const char* type, const std::vector<uint8_t>& v) { put_str(o, name); put_str(o, type); put<uint32_t>(o, (uint32_t)v.size()); o.insert(o.end(), v.begin(), v.end()); }

// host threads this process may use: hardware threads, capped by the affinity mask AND by the cgroup CPU quota (the GPU boxes show 256 logical CPUs
// under a 16-CPU quota: more runnable threads than that are only throttled), at most 32
static unsigned usable_host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (!n) n = 1;
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = std::min<unsigned>(n, (unsigned)CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, atol(q) / period));
        fclose(f);
    } else {
        long quota = -1, period = 0;
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
        if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max(1L, quota / period));
    }
    if (const char* e = getenv("MSNE_HOST_THREADS")) n = (unsigned)std::max(1, atoi(e));   // (1: everything on the calling thread)
    return std::max(1u, std::min(n, 32u));
}

void parallel_for(uint32_t n, const std::function<void(uint32_t)>& job) {
    static const unsigned host_threads = usable_host_threads();
    const unsigned nthreads = std::max(1u, std::min(host_threads, n));
    std::atomic<uint32_t> next{ 0 };
    auto worker = [&] { for (uint32_t i; (i = next.fetch_add(1)) < n;) job(i); };
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    // a thread that cannot be started (resource limits) is not an error: what was started is joined, the calling thread does the rest
    try { for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(worker); } catch (const std::system_error&) {}
    worker();
    for (auto& t : pool) t.join();
}

bool exr_save_rgb(const std::string& path, const float* rgba, uint32_t w, uint32_t h, std::string& err) {
    std::vector<uint8_t> o;
    put<uint32_t>(o, 20000630u); put<uint32_t>(o, 2u);
    { std::vector<uint8_t> v; for (const char* n : { "B", "G", "R" }) { put_str(v, n); put<int32_t>(v, 2); put<uint8_t>(v, 0); put<uint8_t>(v, 0); put<uint8_t>(v, 0); put<uint8_t>(v, 0); put<int32_t>(v, 1); put<int32_t>(v, 1); } put<uint8_t>(v, 0); put_attr(o, "channels", "chlist", v); }
    { std::vector<uint8_t> v; put<uint8_t>(v, 3); put_attr(o, "compression", "compression", v); }   // ZIP
    { std::vector<uint8_t> v; put<int32_t>(v, 0); put<int32_t>(v, 0); put<int32_t>(v, (int32_t)w - 1); put<int32_t>(v, (int32_t)h - 1); put_attr(o, "dataWindow", "box2i", v); put_attr(o, "displayWindow", "box2i", v); }
    { std::vector<uint8_t> v; put<uint8_t>(v, 0); put_attr(o, "lineOrder", "lineOrder", v); }
    { std::vector<uint8_t> v; put<float>(v, 1.0f); put_attr(o, "pixelAspectRatio", "float", v); }
    { std::vector<uint8_t> v; put<float>(v, 0.0f); put<float>(v, 0.0f); put_attr(o, "screenWindowCenter", "v2f", v); }
    { std::vector<uint8_t> v; put<float>(v, 1.0f); put_attr(o, "screenWindowWidth", "float", v); }
    put<uint8_t>(o, 0);
    // ZIP blocks of 16 scanlines, as tinyexr's InitEXRHeader default does for Rgba2D.save: per line the channels B, G, R;
    // bytes split into even/odd halves, delta-predicted, deflated; a block that does not shrink is stored raw
    const size_t line_bytes = (size_t)w * 12, table = o.size();
    const uint32_t nblocks = (h + 15u) / 16u;
    o.resize(table + (size_t)nblocks * 8);
    // The blocks are independent: they are packed on all the host threads this process may use (deflate of a 1080p float film is 0.5 s on one thread,
    // more than rendering it), then appended in order — the file is byte for byte what the one-thread loop writes.
    std::vector<std::vector<uint8_t>> packed_blocks(nblocks);
    auto pack_block = [&](uint32_t b) {
        std::vector<uint8_t> raw, tmp, z;
        const uint32_t y0 = b * 16u, y1 = std::min(h, y0 + 16u);
        raw.resize((size_t)(y1 - y0) * line_bytes);
        for (uint32_t y = y0; y < y1; y++)
            for (int c = 0; c < 3; c++) for (uint32_t x = 0; x < w; x++)
                memcpy(&raw[(size_t)(y - y0) * line_bytes + ((size_t)c * w + x) * 4], &rgba[((size_t)y * w + x) * 4 + (2 - c)], 4);   // B, G, R
        tmp.resize(raw.size());
        { const size_t half = (raw.size() + 1) / 2; size_t a = 0, bq = half; for (size_t i = 0; i < raw.size(); i++) { if (i & 1) tmp[bq++] = raw[i]; else tmp[a++] = raw[i]; } }
        { int prev = tmp.empty() ? 0 : tmp[0]; for (size_t i = 1; i < tmp.size(); i++) { const int cur = tmp[i]; tmp[i] = (uint8_t)(cur - prev + 128 + 256); prev = cur; } }
        uLongf zn = compressBound((uLong)tmp.size());
        z.resize(zn);
        const bool packed = compress(z.data(), &zn, tmp.data(), (uLong)tmp.size()) == Z_OK && zn < raw.size();
        if (packed) { z.resize(zn); packed_blocks[b].swap(z); } else packed_blocks[b].swap(raw);
    };
    parallel_for(nblocks, pack_block);
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint64_t off = o.size();
        memcpy(&o[table + (size_t)b * 8], &off, 8);
        put<int32_t>(o, (int32_t)(b * 16u)); put<int32_t>(o, (int32_t)packed_blocks[b].size());
        o.insert(o.end(), packed_blocks[b].begin(), packed_blocks[b].end());
    }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot write " + path; return false; }
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    fclose(f);
    if (!ok) err = "short write to " + path;
    return ok;
}

}  // namespace msne_host
