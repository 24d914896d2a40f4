// shade.h — device-side restatement of shaders/hrtsystem/{world,material,light}.hlsl (cites per function).
// Expression order is identical to the test oracle's restatement so radiance is bit-exact.
#pragma once
#include "msne_device.h"

namespace msne {

// ---------------- textures: dTextures[i].SampleLevel(dTextureSampler, uv, 0) ----------------
// linear filter, repeat addressing (MaterialManager.zig:429-445); env map: mirrored repeat (BackgroundManager.zig:78-95).
// 1x1 textures (all constant parameters, World.zig:44-228) are returned exactly.
__device__ __forceinline__ int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
__device__ __forceinline__ int wrap_mirror(int i, int n) { int p = 2 * n; int m = i % p; if (m < 0) m += p; return m < n ? m : p - 1 - m; }

__device__ __forceinline__ float4 sample_bilinear(const float4* texels, uint32_t w, uint32_t h, float u, float v, bool mirror) {
    if (w == 1 && h == 1) return texels[0];
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx0 = floor_(x), fy0 = floor_(y);
    const float fx = x - fx0, fy = y - fy0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    int xa, xb, ya, yb;
    if (mirror) { xa = wrap_mirror(x0, (int)w); xb = wrap_mirror(x0 + 1, (int)w); ya = wrap_mirror(y0, (int)h); yb = wrap_mirror(y0 + 1, (int)h); }
    else { xa = wrap_repeat(x0, (int)w); xb = wrap_repeat(x0 + 1, (int)w); ya = wrap_repeat(y0, (int)h); yb = wrap_repeat(y0 + 1, (int)h); }
    const float4 a = texels[(size_t)ya * w + xa], b = texels[(size_t)ya * w + xb], c = texels[(size_t)yb * w + xa], d = texels[(size_t)yb * w + xb];
    float4 o;
    { float top = a.x * (1.0f - fx) + b.x * fx, bot = c.x * (1.0f - fx) + d.x * fx; o.x = top * (1.0f - fy) + bot * fy; }
    { float top = a.y * (1.0f - fx) + b.y * fx, bot = c.y * (1.0f - fx) + d.y * fx; o.y = top * (1.0f - fy) + bot * fy; }
    { float top = a.z * (1.0f - fx) + b.z * fx, bot = c.z * (1.0f - fx) + d.z * fx; o.z = top * (1.0f - fy) + bot * fy; }
    { float top = a.w * (1.0f - fx) + b.w * fx, bot = c.w * (1.0f - fx) + d.w * fx; o.w = top * (1.0f - fy) + bot * fy; }
    return o;
}
// material textures: the same filter over texels decoded from the texture's own format (TexDesc::format) as they are fetched.  The format is
// decided ONCE per lookup and the four texels of a case are loaded together (a switch per texel makes four dependent load -> decode chains of them).
__device__ __forceinline__ float4 sample_bilinear_fmt(const uint4* texels, const float* srgb, uint32_t offset, uint32_t w, uint32_t h, uint32_t fmt, float u, float v) {
    const uint8_t* base = reinterpret_cast<const uint8_t*>(texels + offset);
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx0 = floor_(x), fy0 = floor_(y);
    const float fx = x - fx0, fy = y - fy0;
    const int x0 = (int)fx0, y0 = (int)fy0;
    const int xa = wrap_repeat(x0, (int)w), xb = wrap_repeat(x0 + 1, (int)w), ya = wrap_repeat(y0, (int)h), yb = wrap_repeat(y0 + 1, (int)h);
    const size_t ia = (size_t)ya * w + xa, ib = (size_t)ya * w + xb, ic = (size_t)yb * w + xa, id = (size_t)yb * w + xb;
    float4 a, b, c, d;
    switch (fmt) {
        case TEX_RGBA8_SRGB: { const uint32_t* p = reinterpret_cast<const uint32_t*>(base); const uint32_t pa = p[ia], pb = p[ib], pc = p[ic], pd = p[id];
                               a = decode_rgba8_srgb(pa, srgb); b = decode_rgba8_srgb(pb, srgb); c = decode_rgba8_srgb(pc, srgb); d = decode_rgba8_srgb(pd, srgb); break; }
        case TEX_RG8_UNORM: { const uint16_t* p = reinterpret_cast<const uint16_t*>(base); const uint32_t pa = p[ia], pb = p[ib], pc = p[ic], pd = p[id];
                              a = decode_rg8_unorm(pa); b = decode_rg8_unorm(pb); c = decode_rg8_unorm(pc); d = decode_rg8_unorm(pd); break; }
        case TEX_R8_UNORM: { const uint32_t pa = base[ia], pb = base[ib], pc = base[ic], pd = base[id];
                             a = decode_r8_unorm(pa); b = decode_r8_unorm(pb); c = decode_r8_unorm(pc); d = decode_r8_unorm(pd); break; }
        case TEX_RGBA32F: { const float4* p = reinterpret_cast<const float4*>(base); a = p[ia]; b = p[ib]; c = p[ic]; d = p[id]; break; }
        case TEX_RG32F: { const float2* p = reinterpret_cast<const float2*>(base); const float2 pa = p[ia], pb = p[ib], pc = p[ic], pd = p[id];
                          a = make_float4(pa.x, pa.y, 0.0f, 1.0f); b = make_float4(pb.x, pb.y, 0.0f, 1.0f); c = make_float4(pc.x, pc.y, 0.0f, 1.0f); d = make_float4(pd.x, pd.y, 0.0f, 1.0f); break; }
        case TEX_R32F: { const float* p = reinterpret_cast<const float*>(base); const float pa = p[ia], pb = p[ib], pc = p[ic], pd = p[id];
                         a = make_float4(pa, 0.0f, 0.0f, 1.0f); b = make_float4(pb, 0.0f, 0.0f, 1.0f); c = make_float4(pc, 0.0f, 0.0f, 1.0f); d = make_float4(pd, 0.0f, 0.0f, 1.0f); break; }
        default: { const uint2* p = reinterpret_cast<const uint2*>(base); const uint2 pa = p[ia], pb = p[ib], pc = p[ic], pd = p[id];
                   a = decode_rgba16f(pa.x, pa.y); b = decode_rgba16f(pb.x, pb.y); c = decode_rgba16f(pc.x, pc.y); d = decode_rgba16f(pd.x, pd.y); break; }
    }
    float4 o;
    { float top = a.x * (1.0f - fx) + b.x * fx, bot = c.x * (1.0f - fx) + d.x * fx; o.x = top * (1.0f - fy) + bot * fy; }
    { float top = a.y * (1.0f - fx) + b.y * fx, bot = c.y * (1.0f - fx) + d.y * fx; o.y = top * (1.0f - fy) + bot * fy; }
    { float top = a.z * (1.0f - fx) + b.z * fx, bot = c.z * (1.0f - fx) + d.z * fx; o.z = top * (1.0f - fy) + bot * fy; }
    { float top = a.w * (1.0f - fx) + b.w * fx, bot = c.w * (1.0f - fx) + d.w * fx; o.w = top * (1.0f - fy) + bot * fy; }
    return o;
}
// the lookup from a descriptor that is already in registers: k_shade fetches the (up to five) descriptors of a hit's material
// together, before the first of them is needed, instead of one dependent descriptor -> texel chain after the other
// TEX = false: the caller knows that every texture of the scene is 1x1 (constant material parameters, the glTF factors of World.zig:44-228) — the descriptor's inline
// texel IS the lookup, and the bilinear sampler with its four texels in flight is not even compiled in (k_shade: 25-55 registers less)
template <bool TEX = true>
__device__ __forceinline__ float4 tex_sample_desc(const SceneView& sc, const TexDesc& t, f2 uv) {
    if (!TEX || (t.w == 1 && t.h == 1)) return t.first;
    return sample_bilinear_fmt(sc.texels, sc.srgb_lut, t.offset, t.w, t.h, t.format, uv.x, uv.y);
}
__device__ __forceinline__ float4 tex_sample(const SceneView& sc, uint32_t idx, f2 uv) { const TexDesc t = sc.textures[idx]; return tex_sample_desc(sc, t, uv); }
__device__ __forceinline__ f3 tex_sample_rgb(const SceneView& sc, uint32_t idx, f2 uv) { float4 o = tex_sample(sc, idx, uv); return F3(o.x, o.y, o.z); }

// ---------------- world.hlsl:86-177 ----------------
struct Attrs { f3 position; f2 texcoord; Frame triangleFrame, frame; };

__device__ __forceinline__ void get_tangent_bitangent(f3 p0, f3 p1, f3 p2, f2 t0, f2 t1, f2 t2, f3& tangent, f3& bitangent) {   // :86-100
    const f2 dT02 = F2(t0.x - t2.x, t0.y - t2.y), dT12 = F2(t1.x - t2.x, t1.y - t2.y);
    const f3 dP02 = sub(p0, p2), dP12 = sub(p1, p2);
    const float det = dT02.x * dT12.y - dT02.y * dT12.x;
    if (det == 0.0f) coordinate_system(normalize(cross(sub(p2, p0), sub(p1, p0))), tangent, bitangent);
    else {
        tangent = normalize(divs(sub(scale(dP02, dT12.y), scale(dP12, dT02.y)), det));
        bitangent = normalize(divs(add(scale(dP02, -dT12.x), scale(dP12, dT02.x)), det));
    }
}
__device__ __forceinline__ f3 interp3(f3 b, f3 a0, f3 a1, f3 a2) { return add(add(scale(a0, b.x), scale(a1, b.y)), scale(a2, b.z)); }   // :102-105
__device__ __forceinline__ f2 interp2(f3 b, f2 a0, f2 a1, f2 a2) { return F2(b.x * a0.x + b.y * a1.x + b.z * a2.x, b.x * a0.y + b.y * a1.y + b.z * a2.y); }
__device__ __forceinline__ Frame frame_in_space(const Frame& f, const m34& toMesh) {   // reflection_frame.hlsl:24-30
    Frame o;
    o.n = normalize(m34_mul_transposed(toMesh, f.n));
    o.s = normalize(m34_mul_transposed(toMesh, f.s));
    o.t = normalize(m34_mul_transposed(toMesh, f.t));
    return o;
}
__device__ __forceinline__ f3 ld3(const float* p, uint32_t i) { return F3(p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]); }
__device__ __forceinline__ f2 ld2(const float* p, uint32_t i) { return F2(p[2 * (size_t)i], p[2 * (size_t)i + 1]); }

// The arithmetic of MeshAttributes::lookupAndInterpolate + inWorld (world.hlsl:122-176) once the three vertices' attributes are in hand
// (t = (0,0),(1,0),(1,1) for a mesh without texcoords, :137-141); k_shade and the probe MsneShadeProbe(18) both run this.
__device__ __forceinline__ Attrs mesh_attributes_core(f3 p0, f3 p1, f3 p2, f2 t0, f2 t1, f2 t2, f3 n0, f3 n1, f3 n2, bool has_normals, f3 bary, const m34& toWorld, const m34& toMesh) {
    Attrs a;
    a.position = interp3(bary, p0, p1, p2);
    a.texcoord = interp2(bary, t0, t1, t2);
    get_tangent_bitangent(p0, p1, p2, t0, t1, t2, a.triangleFrame.s, a.triangleFrame.t);
    a.triangleFrame.n = normalize(cross(sub(p0, p2), sub(p1, p2)));
    frame_reorthogonalize(a.triangleFrame);
    if (has_normals) {
        a.frame = a.triangleFrame;
        a.frame.n = normalize(interp3(bary, n0, n1, n2));
        frame_reorthogonalize(a.frame);
    } else a.frame = a.triangleFrame;
    // inWorld
    a.position = m34_mul_point(toWorld, a.position);
    const bool own_frame = has_normals;   // without vertex normals the two frames are the same vectors: the same operations give the same bits
    a.triangleFrame = frame_in_space(a.triangleFrame, toMesh);
    a.frame = own_frame ? frame_in_space(a.frame, toMesh) : a.triangleFrame;
    return a;
}

// MeshAttributes::lookupAndInterpolate(...).inWorld(...)  world.hlsl:114-176; also returns the geometry record.
// Two entry points: by (instance, geometry, primitive) — the reference's chain instance → geometry → mesh → indices →
// positions, used for sampled light triangles — and by triangle-record slot for surface hits (`tri_slot` != MAX_UINT):
// the BVH's 48-B record already holds the three vertices (bit-identical copies), the geometry and the primitive index,
// which removes three dependent loads from every hit, and the 64-B TriAttr in the same slot holds the normals and texcoords the
// builder gathered for it (by vertex index or by corner, as the pipeline reads them) — read only when the mesh has either.
__device__ __forceinline__ Attrs mesh_attributes_world(const SceneView& sc, bool indexed_attributes, uint32_t instanceIndex, uint32_t geometryIndex,
                                                       uint32_t primitiveIndex, f2 attribs, GeometryRec& geo_out, uint32_t tri_slot = MAX_UINT,
                                                       bool geo_known = false /* geo_out already holds the geometry record of the hit */) {
    const InstanceRec* inst = sc.instances + instanceIndex;
    const f3 bary = F3(1.0f - attribs.x - attribs.y, attribs.x, attribs.y);
    f3 p0, p1, p2, n0, n1, n2; f2 t0, t1, t2;
    GeometryRec g; bool has_normals;
    if (tri_slot != MAX_UINT) {
        const uint4* tp = reinterpret_cast<const uint4*>(sc.tris + tri_slot);
        const uint4 ta = tp[0], tb = tp[1], tc = tp[2];
        p0 = F3(u2f(ta.x), u2f(ta.y), u2f(ta.z)); p1 = F3(u2f(ta.w), u2f(tb.x), u2f(tb.y)); p2 = F3(u2f(tb.z), u2f(tb.w), u2f(tc.x));
        geometryIndex = tc.y; primitiveIndex = tc.z;
        g = geo_known ? geo_out : sc.geometries[inst->geo_offset + geometryIndex];
        has_normals = (g.sampled & GEO_HAS_NORMALS) != 0;
        t0 = F2(0.0f, 0.0f); t1 = F2(1.0f, 0.0f); t2 = F2(1.0f, 1.0f);
        n0 = n1 = n2 = F3(0.0f, 0.0f, 0.0f);
        if (g.sampled & (GEO_HAS_TEXCOORDS | GEO_HAS_NORMALS)) {   // the TriAttr in the same slot: the attributes the builder gathered for this triangle
            const float4* ap = reinterpret_cast<const float4*>(sc.tri_attrs + tri_slot);
            const float4 qa = ap[0], qb = ap[1], qc = ap[2], qd = ap[3];
            n0 = F3(qa.x, qa.y, qa.z); n1 = F3(qa.w, qb.x, qb.y); n2 = F3(qb.z, qb.w, qc.x);
            t0 = F2(qc.y, qc.z); t1 = F2(qc.w, qd.x); t2 = F2(qd.y, qd.z);
        }
    } else {
        const uint32_t instanceID = inst->geo_offset;
        g = sc.geometries[instanceID + geometryIndex];
        const MeshRec mesh = sc.meshes[g.mesh];
        const uint32_t i0 = mesh.indices[3 * (size_t)primitiveIndex], i1 = mesh.indices[3 * (size_t)primitiveIndex + 1], i2 = mesh.indices[3 * (size_t)primitiveIndex + 2];
        p0 = ld3(mesh.positions, i0); p1 = ld3(mesh.positions, i1); p2 = ld3(mesh.positions, i2);
        uint32_t a0, a1, a2;
        if (indexed_attributes) { a0 = i0; a1 = i1; a2 = i2; }
        else { a0 = primitiveIndex * 3 + 0; a1 = primitiveIndex * 3 + 1; a2 = primitiveIndex * 3 + 2; }
        if (mesh.texcoords) { t0 = ld2(mesh.texcoords, a0); t1 = ld2(mesh.texcoords, a1); t2 = ld2(mesh.texcoords, a2); }
        else { t0 = F2(0.0f, 0.0f); t1 = F2(1.0f, 0.0f); t2 = F2(1.0f, 1.0f); }
        has_normals = mesh.normals != nullptr;
        n0 = n1 = n2 = F3(0.0f, 0.0f, 0.0f);
        if (has_normals) { n0 = ld3(mesh.normals, a0); n1 = ld3(mesh.normals, a1); n2 = ld3(mesh.normals, a2); }
    }
    geo_out = g;
    return mesh_attributes_core(p0, p1, p2, t0, t1, t2, n0, n1, n2, has_normals, bary, inst->transform, inst->world_to_instance);
}

// ---------------- material.hlsl ----------------
struct MSample { f3 dirFs; float pdf; };
struct Mat { uint32_t type; f3 color; float metalness, alpha, ior; };

constexpr uint32_t MAT_GLASS = 0, MAT_LAMBERT = 1, MAT_MIRROR = 2, MAT_PBR = 3;   // world.hlsl:31-36

__device__ __forceinline__ Mat material_load(const SceneView& sc, const MaterialRec& m, f2 uv) {   // material.hlsl:400-406 + :186-199, :146-152, :348-352
    Mat o; o.type = m.type; o.color = F3(0.0f, 0.0f, 0.0f); o.metalness = 0.0f; o.alpha = 0.0f; o.ior = m.ior;
    if (m.type == MAT_PBR) {
        o.color = tex_sample_rgb(sc, m.color, uv);
        o.metalness = tex_sample(sc, m.metalness, uv).x;
        const float roughness = tex_sample(sc, m.roughness, uv).x;
        o.alpha = maxf(roughness * roughness, 0.001f);
    } else if (m.type == MAT_LAMBERT) o.color = tex_sample_rgb(sc, m.color, uv);
    return o;
}
// ... with the descriptors of color / metalness / roughness already loaded (same lookups, same order)
template <bool TEX = true>
__device__ __forceinline__ Mat material_load_desc(const SceneView& sc, const MaterialRec& m, const TexDesc& tc, const TexDesc& tm, const TexDesc& tr, f2 uv) {
    Mat o; o.type = m.type; o.color = F3(0.0f, 0.0f, 0.0f); o.metalness = 0.0f; o.alpha = 0.0f; o.ior = m.ior;
    if (m.type == MAT_PBR) {
        const float4 c = tex_sample_desc<TEX>(sc, tc, uv); o.color = F3(c.x, c.y, c.z);
        o.metalness = tex_sample_desc<TEX>(sc, tm, uv).x;
        const float roughness = tex_sample_desc<TEX>(sc, tr, uv).x;
        o.alpha = maxf(roughness * roughness, 0.001f);
    } else if (m.type == MAT_LAMBERT) { const float4 c = tex_sample_desc<TEX>(sc, tc, uv); o.color = F3(c.x, c.y, c.z); }
    return o;
}
// GGX :20-67
__device__ __forceinline__ float ggx_D(float alpha, f3 m) { float a2 = alpha * alpha; float t = (m.z * m.z) * (a2 - 1.0f) + 1.0f; float denom = PI * (t * t); return a2 / denom; }
__device__ __forceinline__ float ggx_Lambda(float alpha, f3 v) { float t2 = frame_tan2theta(v); if (isinf_(t2)) return 0.0f; return (sqrt_(1.0f + (alpha * alpha) * t2) - 1.0f) / 2.0f; }
__device__ __forceinline__ float ggx_G(float alpha, f3 wi, f3 wo) { return 1.0f / (1.0f + ggx_Lambda(alpha, wi) + ggx_Lambda(alpha, wo)); }
__device__ __forceinline__ f3 ggx_sample(float alpha, f3 wo, f2 sq) {
    const float tan2 = alpha * alpha * sq.x / (1.0f - sq.x);
    const float cos2 = 1.0f / (1.0f + tan2);
    const float sinT = sqrt_(maxf(0.0f, 1.0f - cos2));
    const float cosT = sqrt_(cos2);
    const float phi = 2.0f * PI * sq.y;
    f3 h = spherical_to_cartesian(sinT, cosT, phi);
    if (!frame_same_hemisphere(wo, h)) h = neg(h);
    return h;
}
__device__ __forceinline__ float ggx_pdf(float alpha, f3 m) { return ggx_D(alpha, m) * absf(m.z); }
// Fresnel :71-123
__device__ __forceinline__ float schlick_weight(float c) { float x = 1.0f - c; float x2 = x * x; return x2 * x2 * x; }
__device__ __forceinline__ float fresnel_dielectric(float cosI, float ei, float et) {
    cosI = clampf(cosI, -1.0f, 1.0f);
    if (!(cosI > 0.0f)) { float tmp = ei; ei = et; et = tmp; cosI = absf(cosI); }
    const float sinI = sqrt_(maxf(0.0f, 1.0f - cosI * cosI));
    const float sinT = ei / et * sinI;
    if (sinT >= 1.0f) return 1.0f;
    const float cosT = sqrt_(maxf(0.0f, 1.0f - sinT * sinT));
    const float r_parl = ((et * cosI) - (ei * cosT)) / ((et * cosI) + (ei * cosT));
    const float r_perp = ((ei * cosI) - (et * cosT)) / ((ei * cosI) + (et * cosT));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
// Lambert :137-175
__device__ __forceinline__ float lambert_pdf(f3 wi, f3 wo) { return frame_same_hemisphere(wi, wo) ? absf(wi.z) / PI : 0.0f; }
__device__ __forceinline__ MSample lambert_sample(f3 wo, f2 sq) {
    f3 wi = square_to_cosine_hemisphere(sq);
    if (wo.z < 0.0f) wi.z *= -1.0f;
    MSample s; s.pdf = lambert_pdf(wi, wo); s.dirFs = wi; return s;
}
// StandardPBR :179-270
__device__ __forceinline__ float pbr_microfacet_pdf(const Mat& m, f3 wi, f3 wo) {
    if (!frame_same_hemisphere(wo, wi)) return 0.0f;
    const f3 h = normalize(add(wi, wo));
    return ggx_pdf(m.alpha, h) / (4.0f * dot(wo, h));
}
__device__ __forceinline__ float pbr_pspec(const Mat& m) { float sw = 1.0f, dw = 1.0f - m.metalness; return sw / (sw + dw); }
__device__ __forceinline__ MSample pbr_sample(const Mat& m, f3 wo, f2 sq) {
    const float pSpec = pbr_pspec(m);
    MSample s;
    if (coin_flip_remap(pSpec, sq.x)) {
        const f3 h = ggx_sample(m.alpha, wo, sq);
        const float k = 2.0f * dot(h, wo);
        const f3 refl = sub(wo, scale(h, k));
        const f3 wi = neg(refl);
        const float mpdf = frame_same_hemisphere(wo, wi) ? ggx_pdf(m.alpha, h) / (4.0f * dot(wo, h)) : 0.0f;
        const float pdf2 = lambert_pdf(wi, wo);
        s.pdf = lerpf(pdf2, mpdf, pSpec); s.dirFs = wi;
    } else {
        const MSample d = lambert_sample(wo, sq);
        const float pdf2 = pbr_microfacet_pdf(m, d.dirFs, wo);
        s.pdf = lerpf(d.pdf, pdf2, pSpec); s.dirFs = d.dirFs;
    }
    return s;
}
__device__ __forceinline__ float pbr_pdf(const Mat& m, f3 wi, f3 wo) { const float pSpec = pbr_pspec(m); return lerpf(lambert_pdf(wi, wo), pbr_microfacet_pdf(m, wi, wo), pSpec); }
__device__ __forceinline__ f3 pbr_eval(const Mat& m, f3 wi, f3 wo) {
    const f3 h = normalize(add(wi, wo));
    const float c = dot(wi, h);
    const float fD = fresnel_dielectric(c, AIR_IOR, m.ior);
    const float w = schlick_weight(c);
    const f3 fM = F3(lerpf(w, 1.0f, m.color.x), lerpf(w, 1.0f, m.color.y), lerpf(w, 1.0f, m.color.z));
    const f3 F = F3(lerpf(fD, fM.x, m.metalness), lerpf(fD, fM.y, m.metalness), lerpf(fD, fM.z, m.metalness));
    const float G = ggx_G(m.alpha, wi, wo);
    const float D = ggx_D(m.alpha, h);
    f3 spec = F3(0.0f, 0.0f, 0.0f);
    if (frame_same_hemisphere(wo, wi)) {
        const float den = 4.0f * absf(wi.z) * absf(wo.z);
        spec = F3(F.x * G * D / den, F.y * G * D / den, F.z * G * D / den);
    }
    const f3 diff = divs(m.color, PI);
    const float k = 1.0f - m.metalness;
    return F3(spec.x + k * diff.x, spec.y + k * diff.y, spec.z + k * diff.z);
}
// Glass :334-393
__device__ __forceinline__ f3 refract_dir(f3 wi, f3 n, float eta) {
    const float cosI = dot(n, wi);
    const float sin2I = maxf(0.0f, 1.0f - cosI * cosI);
    const float sin2T = eta * eta * sin2I;
    if (sin2T >= 1.0f) return F3(0.0f, 0.0f, 0.0f);
    const float cosT = sqrt_(1.0f - sin2T);
    const f3 a = scale(neg(wi), eta);
    const float k = eta * cosI - cosT;
    return add(a, scale(n, k));
}
__device__ __forceinline__ MSample glass_sample(const Mat& m, f3 wo, f2 sq) {
    const float fr = fresnel_dielectric(wo.z, AIR_IOR, m.ior);
    MSample s;
    if (sq.x < fr) { s.pdf = fr; s.dirFs = F3(-wo.x, -wo.y, wo.z); }
    else {
        float ei, et;
        if (wo.z > 0.0f) { ei = AIR_IOR; et = m.ior; } else { et = AIR_IOR; ei = m.ior; }
        s.dirFs = refract_dir(wo, face_forward(F3(0.0f, 0.0f, 1.0f), wo), ei / et);
        s.pdf = (s.dirFs.x == 0.0f && s.dirFs.y == 0.0f && s.dirFs.z == 0.0f) ? 0.0f : 1.0f - fr;
    }
    return s;
}
__device__ __forceinline__ f3 glass_eval(const Mat& m, f3 wi, f3 wo) {
    const float fr = fresnel_dielectric(wo.z, AIR_IOR, m.ior);
    const float e = frame_same_hemisphere(wi, wo) ? fr / absf(wi.z) : (1.0f - fr) / absf(wi.z);
    return F3(e, e, e);
}
// MaterialVariant :395-487
__device__ __forceinline__ bool material_is_delta(const Mat& m) { return m.type == MAT_GLASS || m.type == MAT_MIRROR; }
__device__ __forceinline__ float material_pdf(const Mat& m, f3 wi, f3 wo) {
    if (m.type == MAT_PBR) return pbr_pdf(m, wi, wo);
    if (m.type == MAT_LAMBERT) return lambert_pdf(wi, wo);
    return 0.0f;
}
__device__ __forceinline__ f3 material_eval(const Mat& m, f3 wi, f3 wo) {
    if (m.type == MAT_PBR) return pbr_eval(m, wi, wo);
    if (m.type == MAT_LAMBERT) return divs(m.color, PI);
    if (m.type == MAT_MIRROR) { const float e = 1.0f / absf(wi.z); return F3(e, e, e); }
    return glass_eval(m, wi, wo);
}
__device__ __forceinline__ MSample material_sample(const Mat& m, f3 wo, f2 sq) {
    if (m.type == MAT_PBR) return pbr_sample(m, wo, sq);
    if (m.type == MAT_LAMBERT) return lambert_sample(wo, sq);
    if (m.type == MAT_MIRROR) { MSample s; s.pdf = 1.0f; s.dirFs = F3(-wo.x, -wo.y, wo.z); return s; }
    return glass_sample(m, wo, sq);
}
// material.hlsl:489-522: decodeNormal / tangentNormalToWorld / createTextureFrame on a sampled normal texel (k_shade and MsneShadeProbe(19))
__device__ __forceinline__ Frame texture_frame_from_texel(float4 o, bool two_component, const Frame& tangentFrame) {
    f3 nts;
    if (two_component) {
        const float rx = o.x * 2.0f - 1.0f, ry = o.y * 2.0f - 1.0f;
        const float dd = rx * rx + ry * ry;
        nts = F3(rx, ry, sqrt_(1.0f - clampf(dd, 0.0f, 1.0f)));
    } else nts = F3(o.x, o.y, o.z);
    const f3 nws = normalize(frame_frame_to_world(tangentFrame, nts));
    Frame f = tangentFrame; f.n = nws; frame_reorthogonalize(f);
    return f;
}
template <bool TEX = true>
__device__ __forceinline__ Frame get_texture_frame(const SceneView& sc, const TexDesc& normal_desc, bool two_component, f2 uv, const Frame& tangentFrame) {
    return texture_frame_from_texel(tex_sample_desc<TEX>(sc, normal_desc, uv), two_component, tangentFrame);
}

// ---------------- light.hlsl ----------------
struct LSample { f3 dirWs, radiance; float pdf; };

__device__ __forceinline__ float env_lum_load(const EnvView& e, uint32_t x, uint32_t y, uint32_t level) {   // Texture2D.Load; out of bounds → 0
    const uint32_t s = e.size >> level;
    if (x >= s || y >= s) return 0.0f;
    return e.lum[e.lum_offset[level] + (size_t)y * s + x];
}
__device__ __forceinline__ f3 env_rgb_load(const EnvView& e, uint32_t x, uint32_t y) {
    if (x >= e.size || y >= e.size) return F3(0.0f, 0.0f, 0.0f);
    const float4 p = e.rgb[(size_t)y * e.size + x];
    return F3(p.x, p.y, p.z);
}
// EnvMap::sample light.hlsl:47-80 without its shadow ray (the caller enqueues it when pdf > 0)
// `top`: the quads of the descent's first stored levels — the 1x1, 2x2, 4x4 and 8x8 quad grids, ENV_TOP_QUADS float4 in that order — staged in LDS by the caller
// (k_shade: 1.3 KB of the 5 KB its workgroup has left), or nullptr.  The descent is a chain of dependent fetches, one per level, and every lane of the machine walks
// the same few texels at its top: four of the eight round trips of a 256^2 map are then LDS reads.  Same values, same operations.
constexpr uint32_t ENV_TOP_LEVELS = 4u, ENV_TOP_QUADS = 1u + 4u + 16u + 64u;
__device__ __forceinline__ void env_top_stage(const EnvView& e, float4* top, uint32_t tid, uint32_t nthreads) {   // (the caller synchronises before the first sample)
    if (e.mip_count < 2u) return;
    for (uint32_t i = tid; i < ENV_TOP_QUADS; i += nthreads) {
        const uint32_t k = i < 1u ? 0u : (i < 5u ? 1u : (i < 21u ? 2u : 3u)), first = k == 0u ? 0u : (k == 1u ? 1u : (k == 2u ? 5u : 21u));   // grid k is 2^k quads wide
        if (k + 2u <= e.mip_count) top[i] = e.quads[e.quad_offset[e.mip_count - 2u - k] + (i - first)];
    }
}
__device__ __forceinline__ LSample env_sample_unoccluded(const EnvView& e, f2 rand, const float4* top = nullptr) {
    const uint32_t size = e.size, mipCount = e.mip_count;
    uint32_t ix = 0, iy = 0;
    // One level of the descent reads the 2x2 quad at (2 ix, 2 iy): the last level is the single texel e.top and three reads out of bounds (0);
    // every other level is one float4 of the quad-packed copy.  Same values, same operations, same order as the texel-by-texel form.
    float4 q = make_float4(e.top, 0.0f, 0.0f, 0.0f);
    float chosen = e.top;   // the texel the descent stands on: after level 0 it is luminanceTexture[idx]
    for (uint32_t level = mipCount; level-- > 0;) {
        if (level + 1 != mipCount) {   // (ix, iy) still name the parent texel here
            const uint32_t k = mipCount - 2u - level, w = (size >> level) >> 1;   // the quad grid of this level is w = 2^k quads wide
            if (top && k < ENV_TOP_LEVELS) q = top[(k == 0u ? 0u : (k == 1u ? 1u : (k == 2u ? 5u : 21u))) + iy * w + ix];
            else q = e.quads[e.quad_offset[level] + (size_t)iy * w + ix];
        }
        ix *= 2; iy *= 2;
        const float px = q.x + q.y;
        const float py = q.z + q.w;
        const bool right = coin_flip_remap(py / (px + py), rand.x);
        ix += right ? 1u : 0u;
        const float qx = right ? q.z : q.x;
        const float qy = right ? q.w : q.y;
        const bool down = coin_flip_remap(qy / (qx + qy), rand.y);
        iy += down ? 1u : 0u;
        chosen = down ? qy : qx;
    }
    const float integral = e.top;
    const float discretePdf = chosen * (float)(size * size) / integral;
    const f2 uv = F2(((float)ix + rand.x) / (float)size, ((float)iy + rand.y) / (float)size);
    LSample ls;
    ls.pdf = discretePdf / (4.0f * PI);
    ls.dirWs = square_to_equal_area_sphere(uv);
    ls.radiance = env_rgb_load(e, ix, iy);
    return ls;
}
// EnvMap::eval light.hlsl:83-97
__device__ __forceinline__ void env_eval(const EnvView& e, f3 dirWs, f3& radiance, float& pdf) {
    const uint32_t size = e.size;
    const f2 uv = square_to_equal_area_sphere_inverse(dirWs);
    const float integral = env_lum_load(e, 0, 0, e.mip_count - 1);
    uint32_t ix = (uint32_t)(uv.x * (float)size), iy = (uint32_t)(uv.y * (float)size);
    if (ix > size) ix = size;
    if (iy > size) iy = size;
    const float discretePdf = env_lum_load(e, ix, iy, 0) * (float)(size * size) / integral;
    pdf = discretePdf / (4.0f * PI);
    radiance = env_rgb_load(e, ix, iy);
}
// EnvMap::incomingRadiance light.hlsl:99-102
__device__ __forceinline__ f3 env_incoming_radiance(const EnvView& e, f3 dirWs) {
    const f2 uv = square_to_equal_area_sphere_inverse(dirWs);
    const float4 o = sample_bilinear(e.rgb, e.size, e.size, uv.x, uv.y, true);
    return F3(o.x, o.y, o.z);
}
__device__ __forceinline__ float area_to_solid_angle(f3 pos1, f3 pos2, f3 dir1, f3 dir2) {   // light.hlsl:105-110
    const f3 d = sub(pos1, pos2);
    const float r2 = dot(d, d);
    const float lightCos = dot(neg(dir1), dir2);
    return lightCos > 0.0f ? r2 / lightCos : 0.0f;
}
__device__ __forceinline__ AliasEntry alias_load(const SceneView& sc, uint32_t entryCount, uint32_t i) {   // robust access: out of bounds → zeros
    if (i > entryCount) return AliasEntry{ 0u, 0.0f, 0u, 0u, 0u };
    return sc.alias[i];
}

}  // namespace msne
