/*
 * ORACLE — test infrastructure only (see orc_math.h header).
 * orc_shade.h: restatement of shaders/hrtsystem/{world,material,light,integrator,camera}.hlsl.
 */
#ifndef ORC_SHADE_H
#define ORC_SHADE_H

#include "orc_scene.h"

int orc_closest_hit(const OrcContext *c, v3 o, v3 d, float tmax, orc_hit *h, orc_counters *cnt);
int orc_shadow_hit(const OrcContext *c, v3 o, v3 d, float tmax, orc_counters *cnt);

/* ---------------- textures ---------------- */
/* dTextures[i].SampleLevel(dTextureSampler, uv, 0): linear filter, repeat addressing (MaterialManager.zig:429-445).
 * 1x1 textures (every constant parameter, World.zig:44-228) are returned exactly. */
static inline void tex_fetch(const orc_texture *t, int x, int y, float out[4]) {
    const float *p = &t->rgba[4 * ((size_t)y * t->w + (size_t)x)];
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3];
}
static inline int wrap_repeat(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
static inline int wrap_mirror(int i, int n) { /* VK_SAMPLER_ADDRESS_MODE_MIRRORED_REPEAT */
    int p = 2 * n; int m = i % p; if (m < 0) m += p; return m < n ? m : p - 1 - m;
}
static inline void tex_sample_bilinear(const orc_texture *t, float u, float v, int mirror, float out[4]) {
    if (t->w == 1 && t->h == 1) { tex_fetch(t, 0, 0, out); return; }
    float x = u * (float)t->w - 0.5f, y = v * (float)t->h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float fx = x - fx0, fy = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0;
    int xa, xb, ya, yb;
    if (mirror) { xa = wrap_mirror(x0, (int)t->w); xb = wrap_mirror(x0 + 1, (int)t->w); ya = wrap_mirror(y0, (int)t->h); yb = wrap_mirror(y0 + 1, (int)t->h); }
    else { xa = wrap_repeat(x0, (int)t->w); xb = wrap_repeat(x0 + 1, (int)t->w); ya = wrap_repeat(y0, (int)t->h); yb = wrap_repeat(y0 + 1, (int)t->h); }
    float a[4], b[4], c[4], d[4];
    tex_fetch(t, xa, ya, a); tex_fetch(t, xb, ya, b); tex_fetch(t, xa, yb, c); tex_fetch(t, xb, yb, d);
    for (int k = 0; k < 4; k++) {
        float top = a[k] * (1.0f - fx) + b[k] * fx;
        float bot = c[k] * (1.0f - fx) + d[k] * fx;
        out[k] = top * (1.0f - fy) + bot * fy;
    }
}
static inline v3 tex_sample_rgb(const OrcContext *c, uint32_t idx, v2 uv) {
    float o[4]; tex_sample_bilinear(&c->textures[idx], uv.x, uv.y, 0, o); return V3(o[0], o[1], o[2]);
}

/* ---------------- world.hlsl:86-177 MeshAttributes ---------------- */
typedef struct { v3 position; v2 texcoord; orc_frame triangleFrame, frame; } orc_attrs;

static inline void get_tangent_bitangent(v3 p0, v3 p1, v3 p2, v2 t0, v2 t1, v2 t2, v3 *tangent, v3 *bitangent) { /* world.hlsl:86-100 */
    v2 dT02 = V2(t0.x - t2.x, t0.y - t2.y), dT12 = V2(t1.x - t2.x, t1.y - t2.y);
    v3 dP02 = v3sub(p0, p2), dP12 = v3sub(p1, p2);
    float det = dT02.x * dT12.y - dT02.y * dT12.x;
    if (det == 0.0f) {
        coordinate_system(v3normalize(v3cross(v3sub(p2, p0), v3sub(p1, p0))), tangent, bitangent);
    } else {
        *tangent = v3normalize(v3div(v3sub(v3scale(dP02, dT12.y), v3scale(dP12, dT02.y)), det));
        *bitangent = v3normalize(v3div(v3add(v3scale(dP02, -dT12.x), v3scale(dP12, dT02.x)), det));
    }
}
static inline v3 interp3(v3 b, v3 a0, v3 a1, v3 a2) { return v3add(v3add(v3scale(a0, b.x), v3scale(a1, b.y)), v3scale(a2, b.z)); } /* world.hlsl:102-105 */
static inline v2 interp2(v3 b, v2 a0, v2 a1, v2 a2) { return V2(b.x * a0.x + b.y * a1.x + b.z * a2.x, b.x * a0.y + b.y * a1.y + b.z * a2.y); }

static inline orc_frame frame_in_space(const orc_frame *f, const m34 *toMesh) { /* reflection_frame.hlsl:24-30 */
    orc_frame o;
    o.n = v3normalize(m34_mul_transposed(toMesh, f->n));
    o.s = v3normalize(m34_mul_transposed(toMesh, f->s));
    o.t = v3normalize(m34_mul_transposed(toMesh, f->t));
    return o;
}

/* the arithmetic of lookupAndInterpolate + inWorld (world.hlsl:122-176) on the three vertices' attributes; the renderer and OrcProbeBatch(18) both run this */
static inline orc_attrs mesh_attributes_core(v3 p0, v3 p1, v3 p2, v2 t0, v2 t1, v2 t2, v3 n0, v3 n1, v3 n2, int has_normals, v3 bary, const m34 *toWorld, const m34 *toMesh) {
    orc_attrs a;
    a.position = interp3(bary, p0, p1, p2);
    a.texcoord = interp2(bary, t0, t1, t2);
    get_tangent_bitangent(p0, p1, p2, t0, t1, t2, &a.triangleFrame.s, &a.triangleFrame.t);
    a.triangleFrame.n = v3normalize(v3cross(v3sub(p0, p2), v3sub(p1, p2)));
    frame_reorthogonalize(&a.triangleFrame);
    if (has_normals) {
        a.frame = a.triangleFrame;
        a.frame.n = v3normalize(interp3(bary, n0, n1, n2));
        frame_reorthogonalize(&a.frame);
    } else a.frame = a.triangleFrame;
    /* inWorld */
    a.position = m34_mul_point(toWorld, a.position);
    a.triangleFrame = frame_in_space(&a.triangleFrame, toMesh);
    a.frame = frame_in_space(&a.frame, toMesh);
    return a;
}

/* lookupAndInterpolate(...).inWorld(...) world.hlsl:114-176 */
static inline orc_attrs mesh_attributes_world(const OrcContext *c, uint32_t instanceIndex, uint32_t geometryIndex, uint32_t primitiveIndex, v2 attribs) {
    const orc_instance *inst = &c->instances[instanceIndex];
    uint32_t instanceID = inst->geo_offset;
    const orc_geometry *g = &c->geometries[instanceID + geometryIndex];
    const orc_mesh *mesh = &c->meshes[g->mesh];
    v3 bary = V3(1.0f - attribs.x - attribs.y, attribs.x, attribs.y);
    uint32_t i0 = mesh->indices[3 * primitiveIndex + 0], i1 = mesh->indices[3 * primitiveIndex + 1], i2 = mesh->indices[3 * primitiveIndex + 2];
    v3 p0 = mesh->positions[i0], p1 = mesh->positions[i1], p2 = mesh->positions[i2];
    uint32_t a0, a1, a2;
    if (c->opts.indexed_attributes) { a0 = i0; a1 = i1; a2 = i2; }
    else { a0 = primitiveIndex * 3 + 0; a1 = primitiveIndex * 3 + 1; a2 = primitiveIndex * 3 + 2; }
    v2 t0, t1, t2;
    if (mesh->texcoords) { t0 = mesh->texcoords[a0]; t1 = mesh->texcoords[a1]; t2 = mesh->texcoords[a2]; }
    else { t0 = V2(0, 0); t1 = V2(1, 0); t2 = V2(1, 1); }
    v3 n0 = V3(0, 0, 0), n1 = n0, n2 = n0;
    if (mesh->normals) { n0 = mesh->normals[a0]; n1 = mesh->normals[a1]; n2 = mesh->normals[a2]; }
    return mesh_attributes_core(p0, p1, p2, t0, t1, t2, n0, n1, n2, mesh->normals != NULL, bary, &inst->transform, &inst->world_to_instance);
}

/* ---------------- material.hlsl ---------------- */
typedef struct { v3 dirFs; float pdf; } orc_msample;
typedef struct { uint32_t type; v3 color; float metalness, alpha, ior; } orc_mat; /* loaded MaterialVariant */

static inline orc_mat material_load(const OrcContext *c, uint32_t materialIdx, v2 uv) { /* material.hlsl:400-406 + per-variant load */
    const orc_material *m = &c->materials[materialIdx];
    orc_mat o; o.type = m->type; o.color = V3(0, 0, 0); o.metalness = 0.0f; o.alpha = 0.0f; o.ior = m->ior;
    if (m->type == MSNE_MATERIAL_STANDARD_PBR) {           /* material.hlsl:186-199 */
        o.color = tex_sample_rgb(c, m->color, uv);
        o.metalness = tex_sample_rgb(c, m->metalness, uv).x;
        float roughness = tex_sample_rgb(c, m->roughness, uv).x;
        o.alpha = orc_maxf(roughness * roughness, 0.001f);
    } else if (m->type == MSNE_MATERIAL_LAMBERT) {          /* material.hlsl:146-152 */
        o.color = tex_sample_rgb(c, m->color, uv);
    }
    return o;
}

/* GGX material.hlsl:20-67 */
static inline float ggx_D(float alpha, v3 m) {
    float a2 = alpha * alpha;
    float t = (m.z * m.z) * (a2 - 1.0f) + 1.0f;
    float denom = ORC_PI * (t * t);
    return a2 / denom;
}
static inline float ggx_Lambda(float alpha, v3 v) {
    float t2 = frame_tan2theta(v);
    if (isinf(t2)) return 0.0f;
    return (sqrtf(1.0f + (alpha * alpha) * t2) - 1.0f) / 2.0f;
}
static inline float ggx_G(float alpha, v3 wi, v3 wo) { return 1.0f / (1.0f + ggx_Lambda(alpha, wi) + ggx_Lambda(alpha, wo)); }
static inline v3 ggx_sample(float alpha, v3 wo, v2 sq) {
    float tan2 = alpha * alpha * sq.x / (1.0f - sq.x);
    float cos2 = 1.0f / (1.0f + tan2);
    float sinT = sqrtf(orc_maxf(0.0f, 1.0f - cos2));
    float cosT = sqrtf(cos2);
    float phi = 2.0f * ORC_PI * sq.y;
    v3 h = spherical_to_cartesian(sinT, cosT, phi);
    if (!frame_same_hemisphere(wo, h)) h = v3neg(h);
    return h;
}
static inline float ggx_pdf(float alpha, v3 m) { return ggx_D(alpha, m) * fabsf(m.z); }

/* Fresnel material.hlsl:71-123 */
static inline float schlick_weight(float c) { float x = 1.0f - c; float x2 = x * x; return x2 * x2 * x; }
static inline float fresnel_dielectric(float cosI, float ei, float et) {
    cosI = orc_clampf(cosI, -1.0f, 1.0f);
    if (!(cosI > 0.0f)) { float tmp = ei; ei = et; et = tmp; cosI = fabsf(cosI); }
    float sinI = sqrtf(orc_maxf(0.0f, 1.0f - cosI * cosI));
    float sinT = ei / et * sinI;
    if (sinT >= 1.0f) return 1.0f;
    float cosT = sqrtf(orc_maxf(0.0f, 1.0f - sinT * sinT));
    float r_parl = ((et * cosI) - (ei * cosT)) / ((et * cosI) + (ei * cosT));
    float r_perp = ((ei * cosI) - (et * cosT)) / ((ei * cosI) + (et * cosT));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}

/* Lambert material.hlsl:137-175 */
static inline float lambert_pdf(v3 wi, v3 wo) { return frame_same_hemisphere(wi, wo) ? fabsf(wi.z) / ORC_PI : 0.0f; }
static inline orc_msample lambert_sample(v3 wo, v2 sq) {
    v3 wi = square_to_cosine_hemisphere(sq);
    if (wo.z < 0.0f) wi.z *= -1.0f;
    orc_msample s; s.pdf = lambert_pdf(wi, wo); s.dirFs = wi; return s;
}

/* StandardPBR material.hlsl:179-270 */
static inline float pbr_microfacet_pdf(const orc_mat *m, v3 wi, v3 wo) {
    if (!frame_same_hemisphere(wo, wi)) return 0.0f;
    v3 h = v3normalize(v3add(wi, wo));
    return ggx_pdf(m->alpha, h) / (4.0f * v3dot(wo, h));
}
static inline float pbr_pspec(const orc_mat *m) { float sw = 1.0f, dw = 1.0f - m->metalness; return sw / (sw + dw); }
static inline orc_msample pbr_sample(const orc_mat *m, v3 wo, v2 sq) {
    float pSpec = pbr_pspec(m);
    orc_msample s;
    if (coin_flip_remap(pSpec, &sq.x)) {
        v3 h = ggx_sample(m->alpha, wo, sq);
        float k = 2.0f * v3dot(h, wo);
        v3 refl = v3sub(wo, v3scale(h, k));       /* HLSL reflect(w_o, h) */
        v3 wi = v3neg(refl);
        float mpdf = frame_same_hemisphere(wo, wi) ? ggx_pdf(m->alpha, h) / (4.0f * v3dot(wo, h)) : 0.0f;
        float pdf2 = lambert_pdf(wi, wo);
        s.pdf = orc_lerpf(pdf2, mpdf, pSpec); s.dirFs = wi;
    } else {
        orc_msample d = lambert_sample(wo, sq);
        float pdf2 = pbr_microfacet_pdf(m, d.dirFs, wo);
        s.pdf = orc_lerpf(d.pdf, pdf2, pSpec); s.dirFs = d.dirFs;
    }
    return s;
}
static inline float pbr_pdf(const orc_mat *m, v3 wi, v3 wo) {
    float pSpec = pbr_pspec(m);
    return orc_lerpf(lambert_pdf(wi, wo), pbr_microfacet_pdf(m, wi, wo), pSpec);
}
static inline v3 pbr_eval(const orc_mat *m, v3 wi, v3 wo) {
    v3 h = v3normalize(v3add(wi, wo));
    float c = v3dot(wi, h);
    float fD = fresnel_dielectric(c, ORC_AIR_IOR, m->ior);
    float w = schlick_weight(c);
    v3 fM = V3(orc_lerpf(w, 1.0f, m->color.x), orc_lerpf(w, 1.0f, m->color.y), orc_lerpf(w, 1.0f, m->color.z));
    v3 F = V3(orc_lerpf(fD, fM.x, m->metalness), orc_lerpf(fD, fM.y, m->metalness), orc_lerpf(fD, fM.z, m->metalness));
    float G = ggx_G(m->alpha, wi, wo);
    float D = ggx_D(m->alpha, h);
    v3 spec = V3(0, 0, 0);
    if (frame_same_hemisphere(wo, wi)) {
        float den = 4.0f * fabsf(wi.z) * fabsf(wo.z);
        spec = V3(F.x * G * D / den, F.y * G * D / den, F.z * G * D / den);
    }
    v3 diff = v3div(m->color, ORC_PI);
    float k = 1.0f - m->metalness;
    return V3(spec.x + k * diff.x, spec.y + k * diff.y, spec.z + k * diff.z);
}

/* Glass material.hlsl:334-393 */
static inline v3 refract_dir(v3 wi, v3 n, float eta) {
    float cosI = v3dot(n, wi);
    float sin2I = orc_maxf(0.0f, 1.0f - cosI * cosI);
    float sin2T = eta * eta * sin2I;
    if (sin2T >= 1.0f) return V3(0, 0, 0);
    float cosT = sqrtf(1.0f - sin2T);
    v3 a = v3scale(v3neg(wi), eta);              /* eta * -wi */
    float k = eta * cosI - cosT;
    return v3add(a, v3scale(n, k));
}
static inline orc_msample glass_sample(const orc_mat *m, v3 wo, v2 sq) {
    float fr = fresnel_dielectric(wo.z, ORC_AIR_IOR, m->ior);
    orc_msample s;
    if (sq.x < fr) { s.pdf = fr; s.dirFs = V3(-wo.x, -wo.y, wo.z); }
    else {
        float ei, et;
        if (wo.z > 0.0f) { ei = ORC_AIR_IOR; et = m->ior; } else { et = ORC_AIR_IOR; ei = m->ior; }
        s.dirFs = refract_dir(wo, face_forward(V3(0, 0, 1), wo), ei / et);
        s.pdf = (s.dirFs.x == 0.0f && s.dirFs.y == 0.0f && s.dirFs.z == 0.0f) ? 0.0f : 1.0f - fr;
    }
    return s;
}
static inline v3 glass_eval(const orc_mat *m, v3 wi, v3 wo) {
    float fr = fresnel_dielectric(wo.z, ORC_AIR_IOR, m->ior);
    float e = frame_same_hemisphere(wi, wo) ? fr / fabsf(wi.z) : (1.0f - fr) / fabsf(wi.z);
    return V3(e, e, e);
}

/* MaterialVariant material.hlsl:395-487 */
static inline int material_is_delta(const orc_mat *m) { return m->type == MSNE_MATERIAL_GLASS || m->type == MSNE_MATERIAL_PERFECT_MIRROR; }
static inline float material_pdf(const orc_mat *m, v3 wi, v3 wo) {
    switch (m->type) {
        case MSNE_MATERIAL_STANDARD_PBR: return pbr_pdf(m, wi, wo);
        case MSNE_MATERIAL_LAMBERT: return lambert_pdf(wi, wo);
        default: return 0.0f;
    }
}
static inline v3 material_eval(const orc_mat *m, v3 wi, v3 wo) {
    switch (m->type) {
        case MSNE_MATERIAL_STANDARD_PBR: return pbr_eval(m, wi, wo);
        case MSNE_MATERIAL_LAMBERT: return v3div(m->color, ORC_PI);
        case MSNE_MATERIAL_PERFECT_MIRROR: { float e = 1.0f / fabsf(wi.z); return V3(e, e, e); }
        default: return glass_eval(m, wi, wo);
    }
}
static inline orc_msample material_sample(const orc_mat *m, v3 wo, v2 sq) {
    switch (m->type) {
        case MSNE_MATERIAL_STANDARD_PBR: return pbr_sample(m, wo, sq);
        case MSNE_MATERIAL_LAMBERT: return lambert_sample(wo, sq);
        case MSNE_MATERIAL_PERFECT_MIRROR: { orc_msample s; s.pdf = 1.0f; s.dirFs = V3(-wo.x, -wo.y, wo.z); return s; }
        default: return glass_sample(m, wo, sq);
    }
}

/* material.hlsl:489-522: decodeNormal / tangentNormalToWorld / createTextureFrame on a sampled normal texel (renderer and OrcProbeBatch(19)) */
static inline orc_frame texture_frame_from_texel(const float o[4], int two_component, const orc_frame *tangentFrame) {
    v3 nts;
    if (two_component) {
        float rx = o[0] * 2.0f - 1.0f, ry = o[1] * 2.0f - 1.0f;
        float dd = rx * rx + ry * ry;
        nts = V3(rx, ry, sqrtf(1.0f - orc_clampf(dd, 0.0f, 1.0f)));
    } else nts = V3(o[0], o[1], o[2]);
    v3 nws = v3normalize(frame_frame_to_world(tangentFrame, nts));
    orc_frame f = *tangentFrame; f.n = nws; frame_reorthogonalize(&f);
    return f;
}
static inline orc_frame get_texture_frame(const OrcContext *c, uint32_t materialIdx, v2 uv, const orc_frame *tangentFrame) {
    const orc_material *m = &c->materials[materialIdx];
    float o[4]; tex_sample_bilinear(&c->textures[m->normal], uv.x, uv.y, 0, o);
    return texture_frame_from_texel(o, c->opts.two_component_normal_texture, tangentFrame);
}
static inline v3 get_emissive(const OrcContext *c, uint32_t materialIdx, v2 uv) { return tex_sample_rgb(c, c->materials[materialIdx].emissive, uv); }

/* ---------------- light.hlsl ---------------- */
typedef struct { v3 dirWs, radiance; float pdf; } orc_lsample;

static inline float env_lum_load(const orc_envmap *e, uint32_t x, uint32_t y, uint32_t level) { /* Texture2D.Load, OOB -> 0 */
    uint32_t s = e->size >> level;
    if (x >= s || y >= s) return 0.0f;
    return e->lum[level][(size_t)y * s + x];
}
static inline v3 env_rgb_load(const orc_envmap *e, uint32_t x, uint32_t y) {
    if (x >= e->size || y >= e->size) return V3(0, 0, 0);
    const float *p = &e->rgb[4 * ((size_t)y * e->size + x)];
    return V3(p[0], p[1], p[2]);
}
/* EnvMap::sample light.hlsl:47-80 */
static inline orc_lsample env_sample(const OrcContext *c, v3 positionWs, v3 normalWs, v2 rand, orc_counters *cnt) {
    const orc_envmap *e = &c->env;
    const uint32_t size = e->size, mipCount = e->mip_count;
    uint32_t ix = 0, iy = 0;
    for (uint32_t level = mipCount; level-- > 0;) {
        ix *= 2; iy *= 2;
        float px = env_lum_load(e, ix + 0, iy + 0, level) + env_lum_load(e, ix + 0, iy + 1, level);
        float py = env_lum_load(e, ix + 1, iy + 0, level) + env_lum_load(e, ix + 1, iy + 1, level);
        ix += (uint32_t)coin_flip_remap(py / (px + py), &rand.x);
        float qx = env_lum_load(e, ix + 0, iy + 0, level);
        float qy = env_lum_load(e, ix + 0, iy + 1, level);
        iy += (uint32_t)coin_flip_remap(qy / (qx + qy), &rand.y);
    }
    const float integral = env_lum_load(e, 0, 0, mipCount - 1);
    const float discretePdf = env_lum_load(e, ix, iy, 0) * (float)(size * size) / integral;
    const v2 uv = V2(((float)ix + rand.x) / (float)size, ((float)iy + rand.y) / (float)size);
    orc_lsample ls;
    ls.pdf = discretePdf / (4.0f * ORC_PI);
    ls.dirWs = square_to_equal_area_sphere(uv);
    ls.radiance = env_rgb_load(e, ix, iy);
    if (ls.pdf > 0.0f && orc_shadow_hit(c, offset_along_normal(positionWs, face_forward(normalWs, ls.dirWs)), ls.dirWs, ORC_INFINITY, cnt)) ls.pdf = 0.0f;
    return ls;
}
/* EnvMap::eval light.hlsl:83-97 */
static inline void env_eval(const OrcContext *c, v3 dirWs, v3 *radiance, float *pdf) {
    const orc_envmap *e = &c->env;
    const uint32_t size = e->size;
    const v2 uv = square_to_equal_area_sphere_inverse(dirWs);
    const float integral = env_lum_load(e, 0, 0, e->mip_count - 1);
    uint32_t ix = (uint32_t)(uv.x * (float)size), iy = (uint32_t)(uv.y * (float)size);
    if (ix > size) ix = size;
    if (iy > size) iy = size;
    const float discretePdf = env_lum_load(e, ix, iy, 0) * (float)(size * size) / integral;
    *pdf = discretePdf / (4.0f * ORC_PI);
    *radiance = env_rgb_load(e, ix, iy);
}
/* EnvMap::incomingRadiance light.hlsl:99-102: bilinear, mirrored repeat (BackgroundManager.zig:78-95) */
static inline v3 env_incoming_radiance(const OrcContext *c, v3 dirWs) {
    v2 uv = square_to_equal_area_sphere_inverse(dirWs);
    orc_texture t; t.rgba = c->env.rgb; t.w = t.h = c->env.size;
    float o[4]; tex_sample_bilinear(&t, uv.x, uv.y, 1, o);
    return V3(o[0], o[1], o[2]);
}

static inline float area_to_solid_angle(v3 pos1, v3 pos2, v3 dir1, v3 dir2) { /* light.hlsl:105-110 */
    v3 d = v3sub(pos1, pos2);
    float r2 = v3dot(d, d);
    float lightCos = v3dot(v3neg(dir1), dir2);
    return lightCos > 0.0f ? r2 / lightCos : 0.0f;
}
static inline orc_alias_entry alias_load(const OrcContext *c, uint32_t entryCount, uint32_t i) { /* robust-access OOB -> zeros */
    orc_alias_entry z; memset(&z, 0, sizeof z);
    if (i > entryCount) return z;
    return c->alias[i];
}
/* MeshLights::sample light.hlsl:130-158 (+ sampleAlias mappings.hlsl:114-126) */
static inline orc_lsample mesh_lights_sample(const OrcContext *c, v3 positionWs, v3 triangleNormalDirWs, v2 rand, orc_counters *cnt) {
    orc_lsample ls; ls.pdf = 0.0f; ls.dirWs = V3(0, 0, 0); ls.radiance = V3(0, 0, 0);
    uint32_t entryCount = c->alias[0].alias;
    float sum = c->alias[0].select;
    if (entryCount == 0 || sum == 0.0f) return ls;
    float scaled = rand.x * (float)entryCount;
    uint32_t idx = (uint32_t)scaled;
    rand.x = scaled - floorf(scaled);
    orc_alias_entry en = alias_load(c, entryCount, 1 + idx);
    if (!coin_flip_remap(en.select, &rand.x)) { idx = en.alias; en = alias_load(c, entryCount, 1 + idx); }
    uint32_t instanceID = c->instances[en.instance].geo_offset;
    v2 bary = square_to_triangle(rand);
    orc_attrs at = mesh_attributes_world(c, en.instance, en.geometry, en.primitive, bary);
    ls.radiance = get_emissive(c, c->geometries[instanceID + en.geometry].material, at.texcoord);
    ls.dirWs = v3normalize(v3sub(at.position, positionWs));
    ls.pdf = area_to_solid_angle(at.position, positionWs, ls.dirWs, at.triangleFrame.n) / sum;
    v3 offL = offset_along_normal(at.position, at.triangleFrame.n);
    v3 offS = offset_along_normal(positionWs, face_forward(triangleNormalDirWs, ls.dirWs));
    float tmax = v3length(v3sub(offL, offS));
    if (ls.pdf > 0.0f && orc_shadow_hit(c, offS, v3normalize(v3sub(offL, offS)), tmax, cnt)) ls.pdf = 0.0f;
    return ls;
}

#endif
