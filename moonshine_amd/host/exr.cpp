// exr.cpp — OpenEXR scanline reader/writer for the hot path's I/O contract (engine/fileformats/exr.zig:126-231,
// which wraps syoyo/tinyexr — un-vendored in the reference, deps/tinyexr is empty; this is a from-scratch codec of
// the published OpenEXR 2 file layout).
//   load: RGBA f32, like tinyexr's LoadEXRFromMemory (exr.zig:208-229): channels R,G,B,(A) by name, HALF or FLOAT
//         or UINT pixels, compression NONE / RLE / ZIPS / ZIP, single-part scanline files, any line order.
//   save: three FLOAT channels in header order B,G,R, scanline, ZIP (exr.zig:137-206 with tinyexr's header defaults; alpha is dropped).
#include "host.h"
#include <zlib.h>
#include <cstdio>
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

bool exr_load(const std::string& path, Image& img, std::string& err) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) { err = "cannot read " + path; return false; }
    Reader r{ file.data(), file.size() };
    if (r.u32() != 20000630u) { err = "not an OpenEXR file"; return false; }
    const uint32_t version = r.u32();
    if ((version & 0xff) != 2) { err = "unsupported EXR version"; return false; }
    if (version & 0x200u) { err = "tiled EXR files are not supported"; return false; }
    if (version & 0x1800u) { err = "multi-part / deep EXR files are not supported"; return false; }
    std::vector<Channel> channels; int compression = 0; int32_t dw[4] = { 0, 0, -1, -1 };
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
        r.pos += size;
    }
    const int64_t W = (int64_t)dw[2] - dw[0] + 1, H = (int64_t)dw[3] - dw[1] + 1;
    if (W <= 0 || H <= 0 || W > 65536 || H > 65536 || channels.empty()) { err = "bad EXR header"; return false; }
    for (auto& c : channels) if (c.xs != 1 || c.ys != 1) { err = "subsampled EXR channels are not supported"; return false; }
    int lines_per_block;
    switch (compression) {
        case 0: case 1: case 2: lines_per_block = 1; break;
        case 3: lines_per_block = 16; break;
        default: err = "EXR compression " + std::to_string(compression) + " is not supported (NONE, RLE, ZIPS, ZIP are)"; return false;
    }
    const size_t nblocks = (size_t)((H + lines_per_block - 1) / lines_per_block);
    if (r.pos > file.size() || nblocks > (file.size() - r.pos) / 8) { err = "truncated EXR offset table"; return false; }   // before anything is sized by the header
    std::vector<uint64_t> offsets(nblocks);
    for (auto& o : offsets) o = r.u64();
    if (!r.ok) { err = "truncated EXR offset table"; return false; }
    size_t line_bytes = 0;
    std::vector<size_t> ch_off(channels.size());
    for (size_t c = 0; c < channels.size(); c++) { ch_off[c] = line_bytes; line_bytes += (size_t)W * (channels[c].type == 1 ? 2 : 4); }
    // which channel feeds which of R,G,B,A (tinyexr: by name; a single channel is replicated)
    int src[4] = { -1, -1, -1, -1 };
    for (size_t c = 0; c < channels.size(); c++) {
        const std::string& nm = channels[c].name;
        if (nm == "R") src[0] = (int)c; else if (nm == "G") src[1] = (int)c; else if (nm == "B") src[2] = (int)c; else if (nm == "A") src[3] = (int)c;
    }
    if (src[0] < 0 && src[1] < 0 && src[2] < 0) { if (channels.size() == 1 || channels[0].name == "Y") src[0] = src[1] = src[2] = 0; else { err = "EXR has no R/G/B channels"; return false; } }
    // a header may claim more pixels than the file can hold: deflate expands at most ~1032:1, RLE 64:1
    if ((double)line_bytes * (double)H > (double)file.size() * 1100.0) { err = "EXR data window exceeds the file's data"; return false; }
    img.w = (uint32_t)W; img.h = (uint32_t)H; img.rgba.assign((size_t)W * H * 4, 0.0f);
    for (size_t i = 0; i < (size_t)W * H; i++) img.rgba[4 * i + 3] = 1.0f;
    std::vector<uint8_t> tmp, raw;
    for (size_t b = 0; b < nblocks; b++) {
        if (offsets[b] >= file.size()) { err = "EXR chunk offset outside the file"; return false; }
        Reader c{ file.data(), file.size(), (size_t)offsets[b] };
        const int32_t y0 = c.i32(); const int32_t dsize = c.i32();
        if (!c.ok || dsize < 0 || !c.need((size_t)dsize)) { err = "truncated EXR chunk"; return false; }
        const int64_t row0 = (int64_t)y0 - dw[1];
        if (row0 < 0 || row0 >= H) { err = "EXR chunk outside the data window"; return false; }
        const size_t nlines = (size_t)std::min<int64_t>(lines_per_block, H - row0);
        const size_t expect = nlines * line_bytes;
        const uint8_t* data = c.p + c.pos;
        if ((size_t)dsize == expect) { raw.assign(data, data + expect); }      // stored uncompressed (also when compression did not help)
        else if (compression == 1) { if (!rle_decode(data, (size_t)dsize, tmp, expect)) { err = "bad RLE data"; return false; } undo_predictor_and_interleave(tmp, raw); }
        else if (compression == 2 || compression == 3) { if (!inflate_all(data, (size_t)dsize, tmp, expect)) { err = "bad ZIP data"; return false; } undo_predictor_and_interleave(tmp, raw); }
        else { err = "EXR chunk size mismatch"; return false; }
        for (size_t l = 0; l < nlines; l++) {
            const uint8_t* line = raw.data() + l * line_bytes;
            float* out = &img.rgba[(size_t)(row0 + (int64_t)l) * W * 4];
            for (int k = 0; k < 4; k++) {
                if (src[k] < 0) continue;
                const Channel& ch = channels[(size_t)src[k]];
                const uint8_t* q = line + ch_off[(size_t)src[k]];
                for (int64_t x = 0; x < W; x++) {
                    float v;
                    if (ch.type == 1) { uint16_t hv; memcpy(&hv, q + 2 * x, 2); v = half_to_float(hv); }
                    else if (ch.type == 2) memcpy(&v, q + 4 * x, 4);
                    else { uint32_t uv; memcpy(&uv, q + 4 * x, 4); v = (float)uv; }
                    out[4 * x + k] = v;
                }
            }
        }
    }
    return true;
}

static void put_str(std::vector<uint8_t>& o, const char* s) { o.insert(o.end(), s, s + strlen(s) + 1); }
template <typename T> static void put(std::vector<uint8_t>& o, T v) { const uint8_t* p = (const uint8_t*)&v; o.insert(o.end(), p, p + sizeof(T)); }
static void put_attr(std::vector<uint8_t>& o, const char* name, const char* type, const std::vector<uint8_t>& v) { put_str(o, name); put_str(o, type); put<uint32_t>(o, (uint32_t)v.size()); o.insert(o.end(), v.begin(), v.end()); }

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
    std::vector<uint8_t> raw, tmp, z;
    for (uint32_t b = 0; b < nblocks; b++) {
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
        const uint64_t off = o.size();
        memcpy(&o[table + (size_t)b * 8], &off, 8);
        put<int32_t>(o, (int32_t)y0); put<int32_t>(o, (int32_t)(packed ? zn : raw.size()));
        if (packed) o.insert(o.end(), z.begin(), z.begin() + zn); else o.insert(o.end(), raw.begin(), raw.end());
    }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot write " + path; return false; }
    const bool ok = fwrite(o.data(), 1, o.size(), f) == o.size();
    fclose(f);
    if (!ok) err = "short write to " + path;
    return ok;
}

}  // namespace msne_host
