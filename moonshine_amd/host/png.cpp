// png.cpp — minimal PNG decoder (zlib inflate + scanline unfilter) for glTF-embedded images.  The reference decodes
// them with zigimg and requires 8-bit RGB (World.zig:50-62 `img.pixels.rgb24`); this accepts every
// colour type and bit depth of the PNG specification (gray 1/2/4/8/16, gray+alpha, RGB, RGBA 8/16, palette 1/2/4/8),
// plain or Adam7-interlaced, and always returns 8-bit RGB (alpha dropped, 16-bit samples keep their high byte).
#include "host.h"
#include <zlib.h>
#include <cstring>
#include <cstdlib>

namespace msne_host {

bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    const bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool png_decode(const uint8_t* d, size_t n, Image8& img, std::string& err) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (n < 8 || memcmp(d, sig, 8) != 0) { err = "not a PNG"; return false; }
    size_t pos = 8; uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    // (scratch that lives as long as its thread: a decode allocates megabytes, and mmap / munmap per image serialise the threads of a parallel import on the
    //  process's address-space lock — measured: 112 images on 8 threads took as long as on one)
    static thread_local std::vector<uint8_t> idat, raw;
    std::vector<uint8_t> plte;
    idat.clear();
    while (pos + 12 <= n) {
        const uint32_t len = be32(d + pos); const uint8_t* type = d + pos + 4; const uint8_t* body = d + pos + 8;
        if (pos + 12 + (size_t)len > n) { err = "truncated PNG"; return false; }
        if (!memcmp(type, "IHDR", 4)) { if (len != 13) { err = "bad PNG IHDR"; return false; } w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: err = "bad PNG colour type"; return false; }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8))
                          || ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!w || !h || !depth_ok || interlace > 1) { err = "unsupported PNG header"; return false; }
    // passes: the whole image, or the seven Adam7 sub-images (PNG spec 8.2), each filtered as an image of its own
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
    static const Pass whole = { 0, 0, 1, 1 };
    const Pass* passes = interlace ? adam7 : &whole; const int n_pass = interlace ? 7 : 1;
    const size_t bits = (size_t)ch * depth, bpp = bits >= 8 ? bits / 8 : 1;   // filter distance in bytes
    size_t raw_size = 0;
    for (int k = 0; k < n_pass; k++) {
        const Pass& ps = passes[k];
        if (w <= ps.x0 || h <= ps.y0) continue;
        const size_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        raw_size += ph * (1 + (pw * bits + 7) / 8);
    }
    // a corrupt IHDR may claim more pixels than the IDAT stream can hold (deflate expands at most ~1032:1)
    if (w > 65536u || h > 65536u || (double)raw_size > (double)idat.size() * 1100.0 + 65536.0) { err = "PNG dimensions exceed its data"; return false; }
    raw.resize(raw_size);
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, idat.data(), (uLong)idat.size()) != Z_OK || rl != raw.size()) { err = "PNG inflate failed"; return false; }
    img.w = w; img.h = h; img.rgb.assign((size_t)w * h * 3, 0);
    const size_t step = depth == 16 ? 2 : 1;   // 16-bit: keep the high byte
    const int gray_scale = depth < 8 ? 255 / ((1 << depth) - 1) : 1;   // 1/2/4-bit gray -> 0..255
    size_t at = 0;
    std::vector<uint8_t> rows[2];
    for (int k = 0; k < n_pass; k++) {
        const Pass& ps = passes[k];
        if (w <= ps.x0 || h <= ps.y0) continue;
        const size_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy, stride = (pw * bits + 7) / 8;
        rows[0].assign(stride, 0); rows[1].assign(stride, 0);
        for (size_t j = 0; j < ph; j++) {
            const uint8_t ft = raw[at]; const uint8_t* src = &raw[at + 1]; at += stride + 1;
            uint8_t* cur = rows[j & 1].data(); const uint8_t* up = j ? rows[(j & 1) ^ 1].data() : nullptr;
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
                int v = src[i];
                switch (ft) {
                    case 0: break;
                    case 1: v += a; break;
                    case 2: v += b; break;
                    case 3: v += (a + b) / 2; break;
                    case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                    default: err = "bad PNG filter"; return false;
                }
                cur[i] = (uint8_t)v;
            }
            const size_t y = ps.y0 + j * ps.dy;
            for (size_t i = 0; i < pw; i++) {
                uint8_t* o = &img.rgb[3 * (y * w + ps.x0 + i * ps.dx)];
                uint8_t s0;   // first sample of the pixel (sub-byte depths: packed high bits first)
                if (depth < 8) { const size_t bit = i * depth; s0 = (uint8_t)((cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)); }
                else s0 = cur[i * bpp];
                if (ctype == 2 || ctype == 6) { const uint8_t* p = &cur[i * bpp]; o[0] = p[0]; o[1] = p[step]; o[2] = p[2 * step]; }
                else if (ctype == 3) { const size_t q = (size_t)s0 * 3; if (q + 3 <= plte.size()) { o[0] = plte[q]; o[1] = plte[q + 1]; o[2] = plte[q + 2]; } }
                else { o[0] = o[1] = o[2] = (uint8_t)(s0 * gray_scale); }
            }
        }
    }
    return true;
}

}  // namespace msne_host
