// glb.cpp — binary glTF 2.0 import following the reference's rules (engine/hrtsystem/World.zig:44-363,
// engine/hrtsystem/Camera.zig:26-51; the reference parses with kooparse/zgltf, un-vendored there):
//  * every node with a mesh becomes an instance with its GLOBAL transform, output rows reordered to Z-up
//    (row0 = glTF x-row, row1 = glTF z-row, row2 = glTF y-row — World.zig:341-345);
//  * every primitive is its own mesh + Geometry{mesh, material, sampled = material name starts with "Emitter"} (:262-272);
//  * attributes POSITION / NORMAL / TEXCOORD_0 only, anything else is an error (:298-311); indices u16 in the
//    reference (:274-287) — u8/u16/u32 accepted here (a 1 M-triangle primitive cannot be indexed with u16);
//  * materials (:44-228): normal map PNG → RG8 unorm else constant (0.5,0.5); emissive PNG → RGBA8 sRGB else
//    emissiveFactor*emissiveStrength; transmissionFactor == 1 → Glass(ior); base colour PNG → RGBA8 sRGB else factor;
//    metallicRoughness texture → R = metalness, G = roughness as two R8 textures (the reference's channel choice);
//    without that texture (metallic,roughness) == (0,1) → Lambert, (1,0) → PerfectMirror, else StandardPBR constants;
//  * camera = first node with a camera: origin T·0, forward normalize(T·(0,0,-1)), up T·(0,1,0), vfov = yfov,
//    aperture 0, focus distance 1 (Camera.zig:26-51).
#include "host.h"
#include <cmath>
#include <cstring>
#include <map>
#include <memory>

namespace msne_host {

// ---------------- minimal JSON ----------------
struct JVal {
    enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
    double num = 0; bool b = false; std::string s;
    std::vector<JVal> arr; std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* k) const { if (t != OBJ) return nullptr; for (auto& kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
    double number(const char* k, double d) const { const JVal* v = get(k); return v && v->t == NUM ? v->num : d; }
    // a JSON number that is not finite or does not fit (NaN, 1e300, ...) reads as -1: every caller rejects negative indices, offsets and counts
    int64_t integer(const char* k, int64_t d) const { const JVal* v = get(k); if (!v || v->t != NUM) return d; return (v->num >= -9.0e15 && v->num <= 9.0e15) ? (int64_t)v->num : -1; }
    std::string str(const char* k, const char* d = "") const { const JVal* v = get(k); return v && v->t == STR ? v->s : std::string(d); }
    size_t size() const { return t == ARR ? arr.size() : 0; }
};
struct JParser {
    const char* p; const char* e; bool ok = true; int depth = 0;
    static constexpr int MAX_DEPTH = 64;   // glTF nests 6-7 levels; a chunk of 200 k '[' must not recurse 200 k frames deep
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    JVal parse() {
        JVal v;
        if (++depth > MAX_DEPTH) { ok = false; depth--; return v; }
        v = parse_value();
        depth--;
        return v;
    }
    JVal parse_value() {
        ws(); JVal v;
        if (p >= e) { ok = false; return v; }
        if (*p == '{') {
            v.t = JVal::OBJ; p++; ws();
            if (p < e && *p == '}') { p++; return v; }
            while (ok) {
                ws(); JVal k = parse(); ws();
                if (k.t != JVal::STR || p >= e || *p != ':') { ok = false; break; }
                p++; v.obj.emplace_back(k.s, parse()); ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == '}') { p++; break; }
                ok = false;
            }
        } else if (*p == '[') {
            v.t = JVal::ARR; p++; ws();
            if (p < e && *p == ']') { p++; return v; }
            while (ok) {
                v.arr.push_back(parse()); ws();
                if (p < e && *p == ',') { p++; continue; }
                if (p < e && *p == ']') { p++; break; }
                ok = false;
            }
        } else if (*p == '"') {
            v.t = JVal::STR; p++;
            while (p < e && *p != '"') {
                if (*p == '\\' && p + 1 < e) { p++; char c = *p++; switch (c) { case 'n': v.s += '\n'; break; case 't': v.s += '\t'; break; case 'u': p += 4; v.s += '?'; break; default: v.s += c; } }
                else v.s += *p++;
            }
            if (p < e) p++; else ok = false;
        } else if (!strncmp(p, "true", 4)) { v.t = JVal::BOOL; v.b = true; p += 4; }
        else if (!strncmp(p, "false", 5)) { v.t = JVal::BOOL; p += 5; }
        else if (!strncmp(p, "null", 4)) { p += 4; }
        else {   // the JSON chunk is NUL-terminated by std::string, so strtod cannot run past it
            char* end = nullptr; v.t = JVal::NUM; v.num = strtod(p, &end);
            if (end == p || end > e || !std::isfinite(v.num)) ok = false;
            p = end;
        }
        return v;
    }
};

// ---------------- 4x4 float matrices, column-major like glTF / zgltf ----------------
struct M4 { float m[4][4]; };   // m[col][row]
static M4 m4_identity() { M4 r{}; for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f; return r; }
static M4 m4_mul(const M4& a, const M4& b) { M4 r{}; for (int c = 0; c < 4; c++) for (int rr = 0; rr < 4; rr++) { float s = 0.0f; for (int k = 0; k < 4; k++) s += a.m[k][rr] * b.m[c][k]; r.m[c][rr] = s; } return r; }
static M4 node_local(const JVal& n) {
    if (const JVal* mm = n.get("matrix")) if (mm->size() == 16) { M4 r; for (int i = 0; i < 16; i++) r.m[i / 4][i % 4] = (float)mm->arr[(size_t)i].num; return r; }
    float t[3] = { 0, 0, 0 }, q[4] = { 0, 0, 0, 1 }, s[3] = { 1, 1, 1 };
    if (const JVal* v = n.get("translation")) for (size_t i = 0; i < 3 && i < v->size(); i++) t[i] = (float)v->arr[i].num;
    if (const JVal* v = n.get("rotation")) for (size_t i = 0; i < 4 && i < v->size(); i++) q[i] = (float)v->arr[i].num;
    if (const JVal* v = n.get("scale")) for (size_t i = 0; i < 3 && i < v->size(); i++) s[i] = (float)v->arr[i].num;
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    M4 r = m4_identity();
    r.m[0][0] = (1 - 2 * (y * y + z * z)) * s[0]; r.m[0][1] = (2 * (x * y + z * w)) * s[0]; r.m[0][2] = (2 * (x * z - y * w)) * s[0];
    r.m[1][0] = (2 * (x * y - z * w)) * s[1]; r.m[1][1] = (1 - 2 * (x * x + z * z)) * s[1]; r.m[1][2] = (2 * (y * z + x * w)) * s[1];
    r.m[2][0] = (2 * (x * z + y * w)) * s[2]; r.m[2][1] = (2 * (y * z - x * w)) * s[2]; r.m[2][2] = (1 - 2 * (x * x + y * y)) * s[2];
    r.m[3][0] = t[0]; r.m[3][1] = t[1]; r.m[3][2] = t[2];
    return r;
}
// World.zig:341-345 / Camera.zig:36-40: output rows (x, z, y) of the glTF matrix
static Mat3x4 to_z_up(const M4& g) {
    Mat3x4 o;
    o.x = F32x4{ g.m[0][0], g.m[1][0], g.m[2][0], g.m[3][0] };
    o.y = F32x4{ g.m[0][2], g.m[1][2], g.m[2][2], g.m[3][2] };
    o.z = F32x4{ g.m[0][1], g.m[1][1], g.m[2][1], g.m[3][1] };
    return o;
}

struct Glb {
    JVal j; const uint8_t* bin = nullptr; size_t bin_len = 0;
    // images decoded ahead of the material loop, all at once on the host threads (a scene's PNGs are most of its load time and independent of each other);
    // per image: 0 = not tried, 1 = decoded, 2 = failed with decode_err
    std::vector<Image8> decoded; std::vector<char> decode_state; std::vector<std::string> decode_err;
    const JVal& arr(const char* k) const { static JVal empty; const JVal* v = j.get(k); return v && v->t == JVal::ARR ? *v : empty; }
};

// accessor → tightly packed component arrays (floats or uint32 indices)
static bool read_accessor(const Glb& g, int64_t idx, int want_components, bool as_index, std::vector<float>& f, std::vector<uint32_t>& u, std::string& err) {
    const JVal& accs = g.arr("accessors");
    if (idx < 0 || (size_t)idx >= accs.size()) { err = "bad accessor index"; return false; }
    const JVal& a = accs.arr[(size_t)idx];
    const int64_t bvi = a.integer("bufferView", -1), count = a.integer("count", 0), ctype = a.integer("componentType", 0);
    const std::string type = a.str("type");
    const int comps = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
    if (comps != want_components) { err = "accessor type " + type + " not expected here"; return false; }
    const JVal& bvs = g.arr("bufferViews");
    if (bvi < 0 || (size_t)bvi >= bvs.size()) { err = "accessor without bufferView"; return false; }
    const JVal& bv = bvs.arr[(size_t)bvi];
    const size_t csize = ctype == 5126 || ctype == 5125 ? 4 : (ctype == 5123 || ctype == 5122 ? 2 : 1);
    const size_t elem = csize * (size_t)comps;
    // every number below comes straight from the file: negative, huge or wrapping values are rejected before any arithmetic on them
    const int64_t bv_off = bv.integer("byteOffset", 0), acc_off = a.integer("byteOffset", 0), bv_stride = bv.integer("byteStride", 0);
    if (count < 0 || bv_off < 0 || acc_off < 0 || bv_stride < 0 || (uint64_t)bv_off > g.bin_len || (uint64_t)acc_off > g.bin_len || (uint64_t)bv_stride > g.bin_len + elem) { err = "accessor out of range of the BIN chunk"; return false; }
    const size_t stride = bv_stride ? (size_t)bv_stride : elem;
    const size_t base = (size_t)bv_off + (size_t)acc_off;
    if (count && (base > g.bin_len || elem > g.bin_len - base || (uint64_t)(count - 1) > (g.bin_len - base - elem) / stride)) { err = "accessor out of range of the BIN chunk"; return false; }
    if (as_index) u.reserve((size_t)count * comps); else f.reserve((size_t)count * comps);
    for (int64_t i = 0; i < count; i++) for (int c = 0; c < comps; c++) {
        const uint8_t* p = g.bin + base + (size_t)i * stride + (size_t)c * csize;
        if (as_index) {
            uint32_t v;
            if (ctype == 5125) memcpy(&v, p, 4); else if (ctype == 5123) { uint16_t s; memcpy(&s, p, 2); v = s; } else if (ctype == 5121) v = *p; else { err = "index accessor must be u8/u16/u32"; return false; }
            u.push_back(v);
        } else {
            if (ctype != 5126) { err = "vertex attribute accessors must be f32"; return false; }
            float v; memcpy(&v, p, 4); f.push_back(v);
        }
    }
    return true;
}

static bool image_rgb(const Glb& g, int64_t texture_index, Image8& img, std::string& err) {
    const JVal& texs = g.arr("textures");
    if (texture_index < 0 || (size_t)texture_index >= texs.size()) { err = "bad texture index"; return false; }
    const int64_t src = texs.arr[(size_t)texture_index].integer("source", -1);
    const JVal& imgs = g.arr("images");
    if (src < 0 || (size_t)src >= imgs.size()) { err = "texture without image source"; return false; }
    const JVal& im = imgs.arr[(size_t)src];
    if (im.str("mimeType") != "image/png") { err = "only image/png textures are supported (World.zig:50)"; return false; }
    const int64_t bvi = im.integer("bufferView", -1);
    const JVal& bvs = g.arr("bufferViews");
    if (bvi < 0 || (size_t)bvi >= bvs.size()) { err = "image without bufferView (external URIs are not supported in .glb)"; return false; }
    const int64_t off = bvs.arr[(size_t)bvi].integer("byteOffset", 0), len = bvs.arr[(size_t)bvi].integer("byteLength", 0);
    if (off < 0 || len < 0 || (uint64_t)off > g.bin_len || (uint64_t)len > g.bin_len - (uint64_t)off) { err = "image out of range"; return false; }
    if ((size_t)src < g.decode_state.size() && g.decode_state[(size_t)src] == 1) { img = g.decoded[(size_t)src]; return true; }
    if ((size_t)src < g.decode_state.size() && g.decode_state[(size_t)src] == 2) { err = g.decode_err[(size_t)src]; return false; }
    return png_decode(g.bin + off, (size_t)len, img, err);
}

static int64_t tex_index(const JVal* parent, const char* key) {
    if (!parent) return -1;
    const JVal* t = parent->get(key);
    return t ? t->integer("index", -1) : -1;
}

// World.zig:44-228
static bool import_material(const Glb& g, const JVal& m, const SceneSink& s, MsneMaterialDesc& d, uint32_t& ntex, std::string& err) {
    memset(&d, 0, sizeof d);
    auto chk = [&](int64_t h) -> uint32_t { ntex++; return (uint32_t)h; };
    Image8 img;
    int64_t ti = tex_index(&m, "normalTexture");
    if (ti >= 0) {
        if (!image_rgb(g, ti, img, err)) return false;
        std::vector<uint8_t> rg((size_t)img.w * img.h * 2);
        for (size_t i = 0; i < (size_t)img.w * img.h; i++) { rg[2 * i] = img.rgb[3 * i]; rg[2 * i + 1] = img.rgb[3 * i + 1]; }
        d.normal = chk(s.create_texture(s.ctx, rg.data(), Extent2D{ img.w, img.h }, MSNE_FORMAT_R8G8_UNORM));
    } else d.normal = chk(s.solid2(s.ctx, F32x2{ 0.5f, 0.5f }));
    auto rgba_srgb = [&](int64_t t, uint32_t& out) -> bool {
        if (!image_rgb(g, t, img, err)) return false;
        std::vector<uint8_t> px((size_t)img.w * img.h * 4);
        for (size_t i = 0; i < (size_t)img.w * img.h; i++) { px[4 * i] = img.rgb[3 * i]; px[4 * i + 1] = img.rgb[3 * i + 1]; px[4 * i + 2] = img.rgb[3 * i + 2]; px[4 * i + 3] = 255; }
        out = chk(s.create_texture(s.ctx, px.data(), Extent2D{ img.w, img.h }, MSNE_FORMAT_R8G8B8A8_SRGB));
        return true;
    };
    const JVal* ext = m.get("extensions");
    ti = tex_index(&m, "emissiveTexture");
    if (ti >= 0) { if (!rgba_srgb(ti, d.emissive)) return false; }
    else {
        float ef[3] = { 0, 0, 0 };
        if (const JVal* v = m.get("emissiveFactor")) for (size_t i = 0; i < 3 && i < v->size(); i++) ef[i] = (float)v->arr[i].num;
        float strength = 1.0f;
        if (ext) if (const JVal* es = ext->get("KHR_materials_emissive_strength")) strength = (float)es->number("emissiveStrength", 1.0);
        d.emissive = chk(s.solid3(s.ctx, F32x3{ ef[0] * strength, ef[1] * strength, ef[2] * strength }));
    }
    float ior = 1.5f, transmission = 0.0f;
    if (ext) { if (const JVal* e = ext->get("KHR_materials_ior")) ior = (float)e->number("ior", 1.5); if (const JVal* e = ext->get("KHR_materials_transmission")) transmission = (float)e->number("transmissionFactor", 0.0); }
    d.ior = ior;
    if (transmission == 1.0f) { d.type = MSNE_MATERIAL_GLASS; return true; }
    const JVal* pbr = m.get("pbrMetallicRoughness");
    ti = tex_index(pbr, "baseColorTexture");
    if (ti >= 0) { if (!rgba_srgb(ti, d.color)) return false; }
    else {
        float bc[3] = { 1, 1, 1 };
        if (pbr) if (const JVal* v = pbr->get("baseColorFactor")) for (size_t i = 0; i < 3 && i < v->size(); i++) bc[i] = (float)v->arr[i].num;
        d.color = chk(s.solid3(s.ctx, F32x3{ bc[0], bc[1], bc[2] }));
    }
    const float metallic = pbr ? (float)pbr->number("metallicFactor", 1.0) : 1.0f, roughness = pbr ? (float)pbr->number("roughnessFactor", 1.0) : 1.0f;
    ti = tex_index(pbr, "metallicRoughnessTexture");
    if (ti >= 0) {
        if (!image_rgb(g, ti, img, err)) return false;
        std::vector<uint8_t> rs((size_t)img.w * img.h), gs((size_t)img.w * img.h);
        for (size_t i = 0; i < rs.size(); i++) { rs[i] = img.rgb[3 * i]; gs[i] = img.rgb[3 * i + 1]; }   // World.zig:171-174: R = metalness, G = roughness
        d.metalness = chk(s.create_texture(s.ctx, rs.data(), Extent2D{ img.w, img.h }, MSNE_FORMAT_R8_UNORM));
        d.roughness = chk(s.create_texture(s.ctx, gs.data(), Extent2D{ img.w, img.h }, MSNE_FORMAT_R8_UNORM));
        d.type = MSNE_MATERIAL_STANDARD_PBR;
    } else if (metallic == 0.0f && roughness == 1.0f) d.type = MSNE_MATERIAL_LAMBERT;
    else if (metallic == 1.0f && roughness == 0.0f) d.type = MSNE_MATERIAL_PERFECT_MIRROR;
    else {
        d.metalness = chk(s.solid1(s.ctx, metallic)); d.roughness = chk(s.solid1(s.ctx, roughness));
        d.type = MSNE_MATERIAL_STANDARD_PBR;
    }
    return true;
}

bool glb_import(const std::string& path, const SceneSink& s, GlbSummary& out, std::string& err) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) { err = "cannot read " + path; return false; }
    if (file.size() < 20 || memcmp(file.data(), "glTF", 4) != 0) { err = "not a .glb file"; return false; }
    uint32_t version, total; memcpy(&version, &file[4], 4); memcpy(&total, &file[8], 4);
    if (version != 2 || total > file.size()) { err = "unsupported glTF container version"; return false; }
    Glb g; size_t pos = 12; std::string json;
    while (pos + 8 <= total) {
        uint32_t len, type; memcpy(&len, &file[pos], 4); memcpy(&type, &file[pos + 4], 4);
        if (pos + 8 + (size_t)len > total) { err = "truncated GLB chunk"; return false; }
        if (type == 0x4E4F534Au) json.assign((const char*)&file[pos + 8], len);
        else if (type == 0x004E4942u && !g.bin) { g.bin = &file[pos + 8]; g.bin_len = len; }
        pos += 8 + (size_t)len;
    }
    JParser jp{ json.data(), json.data() + json.size() };
    g.j = jp.parse();
    if (!jp.ok || g.j.t != JVal::OBJ) { err = "GLB JSON chunk does not parse"; return false; }

    {   // decode every PNG some texture refers to, in parallel; what image_rgb checks (mime type, ranges) stays with image_rgb, which also reports the errors
        const JVal& imgs = g.arr("images"); const JVal& bvs = g.arr("bufferViews");
        std::vector<uint32_t> todo; std::vector<char> wanted(imgs.size(), 0);
        for (const JVal& t : g.arr("textures").arr) { const int64_t src = t.integer("source", -1); if (src >= 0 && (size_t)src < imgs.size()) wanted[(size_t)src] = 1; }
        g.decoded.resize(imgs.size()); g.decode_state.assign(imgs.size(), 0); g.decode_err.resize(imgs.size());
        std::vector<std::pair<size_t, size_t>> range(imgs.size());
        for (size_t i = 0; i < imgs.size(); i++) {
            if (!wanted[i] || imgs.arr[i].str("mimeType") != "image/png") continue;
            const int64_t bvi = imgs.arr[i].integer("bufferView", -1);
            if (bvi < 0 || (size_t)bvi >= bvs.size()) continue;
            const int64_t off = bvs.arr[(size_t)bvi].integer("byteOffset", 0), len = bvs.arr[(size_t)bvi].integer("byteLength", 0);
            if (off < 0 || len < 0 || (uint64_t)off > g.bin_len || (uint64_t)len > g.bin_len - (uint64_t)off) continue;
            range[i] = { (size_t)off, (size_t)len }; todo.push_back((uint32_t)i);
        }
        parallel_for((uint32_t)todo.size(), [&](uint32_t k) {
            const size_t i = todo[k];
            g.decode_state[i] = png_decode(g.bin + range[i].first, range[i].second, g.decoded[i], g.decode_err[i]) ? 1 : 2;
        });
    }

    // materials (World.zig:234-248)
    const JVal& mats = g.arr("materials");
    std::vector<uint32_t> mat_handle; std::vector<std::string> mat_name;
    for (const JVal& m : mats.arr) {
        MsneMaterialDesc d;
        if (!import_material(g, m, s, d, out.textures, err)) return false;
        const int64_t h = s.create_material(s.ctx, &d);
        if (h < 0) { err = "material rejected"; return false; }
        mat_handle.push_back((uint32_t)h); mat_name.push_back(m.str("name"));
        out.materials++;
    }
    int64_t default_mat = -1;

    // global transforms: parent chain (zgltf getGlobalTransform)
    const JVal& nodes = g.arr("nodes");
    std::vector<int64_t> parent(nodes.size(), -1);
    for (size_t i = 0; i < nodes.size(); i++) if (const JVal* ch = nodes.arr[i].get("children")) for (const JVal& c : ch->arr) if (c.t == JVal::NUM && c.num >= 0.0 && c.num < (double)nodes.size()) parent[(size_t)c.num] = (int64_t)i;
    auto global = [&](size_t i) { M4 m = node_local(nodes.arr[i]); int guard = 0; for (int64_t p = parent[i]; p >= 0 && guard < 1024; p = parent[(size_t)p], guard++) m = m4_mul(node_local(nodes.arr[(size_t)p]), m); return m; };

    const JVal& meshes = g.arr("meshes");
    for (size_t ni = 0; ni < nodes.size(); ni++) {
        const int64_t mi = nodes.arr[ni].integer("mesh", -1);
        if (mi < 0) continue;
        if ((size_t)mi >= meshes.size()) { err = "node references a missing mesh"; return false; }
        const JVal* prims = meshes.arr[(size_t)mi].get("primitives");
        std::vector<Geometry> geos;
        for (size_t pi = 0; prims && pi < prims->size(); pi++) {
            const JVal& pr = prims->arr[pi];
            if (pr.integer("mode", 4) != 4) { err = "only triangle lists are supported"; return false; }
            std::vector<float> pos3, nrm, uv; std::vector<uint32_t> idx, dummy_u; std::vector<float> dummy_f;
            const JVal* attrs = pr.get("attributes");
            if (!attrs || attrs->t != JVal::OBJ) { err = "primitive without attributes"; return false; }
            for (auto& kv : attrs->obj) {
                const int64_t a = (kv.second.t == JVal::NUM && kv.second.num >= 0.0 && kv.second.num < 4.0e9) ? (int64_t)kv.second.num : -1;
                if (kv.first == "POSITION") { if (!read_accessor(g, a, 3, false, pos3, dummy_u, err)) return false; }
                else if (kv.first == "NORMAL") { if (!read_accessor(g, a, 3, false, nrm, dummy_u, err)) return false; }
                else if (kv.first == "TEXCOORD_0") { if (!read_accessor(g, a, 2, false, uv, dummy_u, err)) return false; }
                else { err = "unhandled vertex attribute " + kv.first + " (World.zig:308-311)"; return false; }
            }
            if (pos3.empty()) { err = "primitive without POSITION"; return false; }
            const int64_t ia = pr.integer("indices", -1);
            if (ia >= 0) { if (!read_accessor(g, ia, 1, true, dummy_f, idx, err)) return false; }
            else for (uint32_t k = 0; k < pos3.size() / 3; k++) idx.push_back(k);
            const size_t nv = pos3.size() / 3, nt = idx.size() / 3;
            if ((!nrm.empty() && nrm.size() / 3 != nv) || (!uv.empty() && uv.size() / 2 != nv)) { err = "attribute counts differ"; return false; }
            const int64_t mh = s.create_mesh(s.ctx, (const F32x3*)pos3.data(), nrm.empty() ? nullptr : (const F32x3*)nrm.data(), uv.empty() ? nullptr : (const F32x2*)uv.data(), nv, nv, (const U32x3*)idx.data(), nt);
            if (mh < 0) { err = "mesh rejected"; return false; }
            out.meshes++; out.triangles += (uint32_t)nt;
            int64_t pm = pr.integer("material", -1);
            Geometry ge; ge.mesh = (MeshHandle)mh; ge.sampled = false;
            if (pm >= 0 && (size_t)pm < mat_handle.size()) { ge.material = mat_handle[(size_t)pm]; ge.sampled = mat_name[(size_t)pm].rfind("Emitter", 0) == 0; }   // World.zig:270
            else {
                if (default_mat < 0) {   // the reference would fail on `primitive.material.?`; glTF's default material instead
                    MsneMaterialDesc d{}; d.normal = (uint32_t)s.solid2(s.ctx, F32x2{ 0.5f, 0.5f }); d.emissive = (uint32_t)s.solid3(s.ctx, F32x3{ 0, 0, 0 });
                    d.color = (uint32_t)s.solid3(s.ctx, F32x3{ 1, 1, 1 }); d.metalness = (uint32_t)s.solid1(s.ctx, 1.0f); d.roughness = (uint32_t)s.solid1(s.ctx, 1.0f);
                    d.type = MSNE_MATERIAL_STANDARD_PBR; d.ior = 1.5f; out.textures += 5;
                    default_mat = s.create_material(s.ctx, &d); out.materials++;
                }
                ge.material = (MaterialHandle)default_mat;
            }
            geos.push_back(ge);
        }
        if (geos.empty()) continue;
        if (s.create_instance(s.ctx, to_z_up(global(ni)), geos.data(), geos.size(), true) < 0) { err = "instance rejected"; return false; }
        out.instances++;
    }

    // Camera.zig:26-51
    const JVal& cams = g.arr("cameras");
    for (size_t ni = 0; ni < nodes.size(); ni++) {
        const int64_t ci = nodes.arr[ni].integer("camera", -1);
        if (ci < 0 || (size_t)ci >= cams.size()) continue;
        const JVal* persp = cams.arr[(size_t)ci].get("perspective");
        if (!persp) { err = "only perspective cameras are supported"; return false; }
        const Mat3x4 T = to_z_up(global(ni));
        auto mul_point = [&](float x, float y, float z) { return F32x3{ T.x.x * x + T.x.y * y + T.x.z * z + T.x.w * 1.0f, T.y.x * x + T.y.y * y + T.y.z * z + T.y.w * 1.0f, T.z.x * x + T.z.y * y + T.z.z * z + T.z.w * 1.0f }; };
        auto mul_vec = [&](float x, float y, float z) { return F32x3{ T.x.x * x + T.x.y * y + T.x.z * z, T.y.x * x + T.y.y * y + T.y.z * z, T.z.x * x + T.z.y * y + T.z.z * z }; };
        Lens l;
        l.origin = mul_point(0, 0, 0);
        F32x3 f = mul_vec(0, 0, -1); const float fl = sqrtf(f.x * f.x + f.y * f.y + f.z * f.z);
        l.forward = F32x3{ f.x / fl, f.y / fl, f.z / fl };
        l.up = mul_vec(0, 1, 0);
        l.vfov = (float)persp->number("yfov", 0.8); l.aperture = 0.0f; l.focus_distance = 1.0f;
        out.lens = s.create_lens(s.ctx, l);
        break;
    }
    if (out.lens < 0) { err = "no camera in the GLB (Camera.zig:30: error.NoCameraInGlb)"; return false; }
    return true;
}

}  // namespace msne_host
