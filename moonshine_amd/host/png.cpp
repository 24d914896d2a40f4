// png.cpp — minimal PNG decoder (zlib inflate + scanline unfilter) for glTF-embedded images.  The reference decodes
// them with zigimg and requires 8-bit RGB (World.zig:50-62 `img.pixels.rgb24`); this accepts 8/16-bit
// gray, gray+alpha, RGB, RGBA and palette images (non-interlaced) and always returns 8-bit RGB.
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
    std::vector<uint8_t> idat, plte;
    while (pos + 12 <= n) {
        const uint32_t len = be32(d + pos); const uint8_t* type = d + pos + 4; const uint8_t* body = d + pos + 8;
        if (pos + 12 + (size_t)len > n) { err = "truncated PNG"; return false; }
        if (!memcmp(type, "IHDR", 4)) { w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!w || !h || (depth != 8 && depth != 16) || interlace) { err = "unsupported PNG (need non-interlaced 8/16-bit)"; return false; }
    int ch;
    switch (ctype) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: err = "bad PNG colour type"; return false; }
    if (ctype == 3 && depth != 8) { err = "unsupported palette depth"; return false; }
    const size_t bpp = (size_t)ch * depth / 8, stride = (size_t)w * bpp;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, idat.data(), (uLong)idat.size()) != Z_OK || rl != raw.size()) { err = "PNG inflate failed"; return false; }
    std::vector<uint8_t> pix(stride * h);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t ft = raw[(stride + 1) * y]; const uint8_t* src = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &pix[stride * y]; const uint8_t* up = y ? &pix[stride * (y - 1)] : nullptr;
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
    }
    img.w = w; img.h = h; img.rgb.resize((size_t)w * h * 3);
    const size_t step = depth / 8;   // 16-bit: keep the high byte
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t* p = &pix[i * bpp]; uint8_t* o = &img.rgb[3 * i];
        if (ctype == 2 || ctype == 6) { o[0] = p[0]; o[1] = p[step]; o[2] = p[2 * step]; }
        else if (ctype == 3) { const size_t k = (size_t)p[0] * 3; if (k + 3 <= plte.size()) { o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2]; } else { o[0] = o[1] = o[2] = 0; } }
        else { o[0] = o[1] = o[2] = p[0]; }
    }
    return true;
}

}  // namespace msne_host
