/*
 * ORACLE — test infrastructure only (see orc_math.h header).  Parity pin: the reference's own
 * furnace tests (engine/tests.zig:257-455) run on this code at the reference's parameters and
 * tolerances (tests/test_oracle_furnace.py); everything the reference does not test is pinned
 * only to SURVEY.md Appendix A ("parity unpinned" for GGX / glass / mirror / mesh lights /
 * non-constant env maps / textures — see DESIGN.md).
 *
 * orc_core.c: integrator (integrator.hlsl:68-183), raygen/film (main.hlsl:43-95), camera
 * (camera.hlsl:14-42), env preprocessing (shaders/background/*.hlsl, BackgroundManager.zig:142-394),
 * alias table (alias_table.zig:25-92, Accel.zig:491-539), and the Orc* C API used through ctypes.
 */
#include <stdlib.h>
#include <stdio.h>
#include <pthread.h>
#include <time.h>
#include "orc_shade.h"

void orc_bvh_free(orc_bvh *b);
void orc_build_blas(OrcContext *c, orc_bvh *out, const uint32_t *mesh_ids, uint32_t ngeo);
void orc_build_tlas(OrcContext *c);

/* ---------------- integrator.hlsl ---------------- */
static inline float power_heuristic(uint32_t numf, float fPdf, uint32_t numg, float gPdf) { /* integrator.hlsl:10-16 */
    float f = (float)numf * fPdf, g = (float)numg * gPdf;
    float f2 = f * f;
    return f2 / (f2 + g * g);
}

/* estimateDirectMISLight integrator.hlsl:20-35 (light already sampled) */
static inline v3 estimate_direct_mis(const orc_frame *frame, const orc_lsample *ls, const orc_mat *mat, v3 woFs, uint32_t samplesTaken) {
    if (ls->pdf > 0.0f) {
        v3 wiFs = frame_world_to_frame(frame, ls->dirWs);
        float scatteringPdf = material_pdf(mat, wiFs, woFs);
        if (scatteringPdf > 0.0f) {
            v3 brdf = material_eval(mat, wiFs, woFs);
            float weight = power_heuristic(samplesTaken, ls->pdf, 1, scatteringPdf);
            float ac = fabsf(wiFs.z);
            return V3(ls->radiance.x * brdf.x * ac * weight / ls->pdf,
                      ls->radiance.y * brdf.y * ac * weight / ls->pdf,
                      ls->radiance.z * brdf.z * ac * weight / ls->pdf);
        }
    }
    return V3(0, 0, 0);
}

/* optional per-path trace (debugging / path-level parity tests): 8 floats per surface hit */
static __thread float *g_dbg = NULL; static __thread uint32_t g_dbg_n = 0, g_dbg_cap = 0;

/* PathTracingIntegrator::incomingRadiance integrator.hlsl:68-183 */
static v3 incoming_radiance(const OrcContext *c, v3 rayO, v3 rayD, orc_rng *rng, orc_counters *cnt) {
    const uint32_t max_bounces = c->opts.max_bounces, env_n = c->opts.env_samples_per_bounce, mesh_n = c->opts.mesh_samples_per_bounce;
    v3 L = V3(0, 0, 0), throughput = V3(1, 1, 1);
    uint32_t bounceCount = 0;
    float lastMaterialPdf = 0.0f;
    int isLastMaterialDelta = 0;
    const float rayTMax = ORC_INFINITY;
    orc_hit its;
    while (orc_closest_hit(c, rayO, rayD, rayTMax, &its, cnt)) {
        cnt->surface_hits++;
        if (g_dbg && g_dbg_n < g_dbg_cap) { float *r = &g_dbg[12 * g_dbg_n++]; r[0] = (float)its.inst; r[1] = (float)its.prim; r[2] = its.t; r[3] = its.u; r[4] = its.v; r[5] = rayD.x; r[6] = rayD.y; r[7] = rayD.z; r[8] = rayO.x; r[9] = rayO.y; r[10] = rayO.z; r[11] = (float)its.geo; }
        uint32_t instanceID = c->instances[its.inst].geo_offset;
        const orc_geometry *geometry = &c->geometries[instanceID + its.geo];
        orc_attrs attrs = mesh_attributes_world(c, its.inst, its.geo, its.prim, V2(its.u, its.v));
        orc_frame textureFrame = get_texture_frame(c, geometry->material, attrs.texcoord, &attrs.frame);
        v3 emissiveLight = get_emissive(c, geometry->material, attrs.texcoord);
        orc_mat material = material_load(c, geometry->material, attrs.texcoord);

        v3 woWs = v3neg(rayD);
        int frontfacing = v3dot(attrs.triangleFrame.n, woWs) > 0.0f;
        orc_frame shadingFrame;
        if ((frontfacing && v3dot(woWs, textureFrame.n) > 0.0f) || (!frontfacing && -v3dot(woWs, textureFrame.n) > 0.0f)) shadingFrame = textureFrame;
        else if ((frontfacing && v3dot(woWs, attrs.frame.n) > 0.0f) || (!frontfacing && -v3dot(woWs, attrs.frame.n) > 0.0f)) shadingFrame = attrs.frame;
        else shadingFrame = attrs.triangleFrame;
        v3 woSs = frame_world_to_frame(&shadingFrame, woWs);

        if (mesh_n == 0 || bounceCount == 0 || !geometry->sampled || isLastMaterialDelta) {
            if (v3dot(woWs, attrs.triangleFrame.n) > 0.0f) L = v3add(L, v3mul(throughput, emissiveLight));
        } else if (geometry->sampled) {
            float sum = c->alias[0].select;
            float lightPdf = area_to_solid_angle(attrs.position, rayO, rayD, attrs.triangleFrame.n) / sum;
            if (lightPdf > 0.0f) {
                float weight = power_heuristic(1, lastMaterialPdf, mesh_n, lightPdf);
                L = v3add(L, v3scale(v3mul(throughput, emissiveLight), weight));
            }
        }

        if (bounceCount >= max_bounces + 1) return L;
        else if (bounceCount > 3) {
            float pSurvive = orc_minf(0.95f, orc_luminance(throughput));
            if (rng_get_float(rng) > pSurvive) return L;
            throughput = v3div(throughput, pSurvive);
        }

        int isCurrentMaterialDelta = material_is_delta(&material);
        if (!isCurrentMaterialDelta) {
            for (uint32_t k = 0; k < env_n; k++) {
                v2 rand; rand.x = rng_get_float(rng); rand.y = rng_get_float(rng);
                orc_lsample ls = env_sample(c, attrs.position, attrs.triangleFrame.n, rand, cnt);
                v3 e = estimate_direct_mis(&shadingFrame, &ls, &material, woSs, env_n);
                L = v3add(L, v3div(v3mul(throughput, e), (float)env_n));
            }
            for (uint32_t k = 0; k < mesh_n; k++) {
                v2 rand; rand.x = rng_get_float(rng); rand.y = rng_get_float(rng);
                orc_lsample ls = mesh_lights_sample(c, attrs.position, attrs.triangleFrame.n, rand, cnt);
                v3 e = estimate_direct_mis(&shadingFrame, &ls, &material, woSs, mesh_n);
                L = v3add(L, v3div(v3mul(throughput, e), (float)mesh_n));
            }
        }

        v2 sq; sq.x = rng_get_float(rng); sq.y = rng_get_float(rng);
        orc_msample sample = material_sample(&material, woSs, sq);
        if (sample.pdf == 0.0f) return L;
        lastMaterialPdf = sample.pdf;

        rayD = frame_frame_to_world(&shadingFrame, sample.dirFs);
        rayO = offset_along_normal(attrs.position, face_forward(attrs.triangleFrame.n, rayD));
        v3 f = material_eval(&material, sample.dirFs, woSs);
        float ac = fabsf(sample.dirFs.z);
        throughput = v3mul(throughput, V3(f.x * ac / sample.pdf, f.y * ac / sample.pdf, f.z * ac / sample.pdf));
        bounceCount += 1;
        isLastMaterialDelta = isCurrentMaterialDelta;
    }
    if (env_n == 0 || bounceCount == 0 || isLastMaterialDelta) {
        L = v3add(L, v3mul(throughput, env_incoming_radiance(c, rayD)));
    } else {
        v3 rad; float pdf;
        env_eval(c, rayD, &rad, &pdf);
        if (pdf > 0.0f) {
            float weight = power_heuristic(1, lastMaterialPdf, env_n, pdf);
            L = v3add(L, v3scale(v3mul(throughput, rad), weight));
        }
    }
    return L;
}

/* Camera::generateRay camera.hlsl:14-42 */
static void generate_ray(const Lens *lens, uint32_t W, uint32_t H, v2 uv, v2 rand, v3 *O, v3 *D) {
    v3 origin = V3(lens->origin.x, lens->origin.y, lens->origin.z);
    v3 forward = V3(lens->forward.x, lens->forward.y, lens->forward.z);
    v3 up = V3(lens->up.x, lens->up.y, lens->up.z);
    float aspect = (float)W / (float)H;
    v3 w = v3scale(forward, -1.0f);
    v3 u = v3normalize(v3cross(up, w));
    v3 v = v3cross(w, u);
    float h = det_tanf(lens->vfov / 2.0f);
    float viewport_height = 2.0f * h * lens->focus_distance;
    float viewport_width = aspect * viewport_height;
    v3 horizontal = v3scale(u, viewport_width);
    v3 vertical = v3scale(v, viewport_height);
    v3 llc = v3sub(v3sub(v3sub(origin, v3div(horizontal, 2.0f)), v3div(vertical, 2.0f)), v3scale(w, lens->focus_distance));
    v2 sr = square_to_uniform_disk_concentric(rand);
    v2 rd = V2(lens->aperture * sr.x / 2.0f, lens->aperture * sr.y / 2.0f);
    v3 defocus = v3add(v3scale(u, rd.x), v3scale(v, rd.y));
    *O = v3add(origin, defocus);
    *D = v3normalize(v3sub(v3sub(v3add(v3add(llc, v3scale(horizontal, uv.x)), v3scale(vertical, uv.y)), defocus), origin));
}

/* raygen + dispatchUV + storeColor main.hlsl:43-95 for one pixel */
static void render_pixel(const OrcContext *c, orc_sensor *s, const Lens *lens, uint32_t x, uint32_t y, orc_counters *cnt) {
    const uint32_t spr = c->opts.samples_per_run;
    v3 color = V3(0, 0, 0);
    for (uint32_t k = 0; k < spr; k++) {
        orc_rng rng = rng_from_seed(s->sample_count + k, x, y);
        v2 r1; r1.x = rng_get_float(&rng); r1.y = rng_get_float(&rng);
        v2 g = square_to_gaussian(r1);
        v2 center = V2(0.5f + 0.5f * g.x, 0.5f + 0.5f * g.y);
        v2 uv = V2(((float)x + center.x) / (float)s->w, ((float)y + center.y) / (float)s->h);
        if (c->opts.flip_image) uv.y = 1.0f - uv.y;
        v2 r2; r2.x = rng_get_float(&rng); r2.y = rng_get_float(&rng);
        v3 O, D; generate_ray(lens, s->w, s->h, uv, r2, &O, &D);
        cnt->samples++;
        color = v3add(color, incoming_radiance(c, O, D, &rng, cnt));
    }
    float *px = &s->film[4 * ((size_t)y * s->w + x)];
    if (s->sample_count == 0) {
        px[0] = color.x / (float)spr; px[1] = color.y / (float)spr; px[2] = color.z / (float)spr; px[3] = 1.0f;
    } else {
        float den = (float)(s->sample_count + spr);
        float p0 = px[0], p1 = px[1], p2 = px[2];
        px[0] = p0 + (color.x - p0) / den; px[1] = p1 + (color.y - p1) / den; px[2] = p2 + (color.z - p2) / den; px[3] = px[3] + 1.0f;
    }
}

/* ---------------- textures ---------------- */
static float half_to_float(uint16_t h) {
    uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 0x1f, m = h & 0x3ff;
    if (e == 0) { if (m == 0) return u2f(s); float f = (float)m * 0x1p-24f; return (h >> 15) ? -f : f; }
    if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
    return u2f(s | ((e + 112) << 23) | (m << 13));
}
static uint32_t add_texture(OrcContext *c, const void *bytes, uint32_t w, uint32_t h, int fmt) {
    c->textures = (orc_texture *)realloc(c->textures, sizeof(orc_texture) * (c->texture_count + 1));
    orc_texture *t = &c->textures[c->texture_count];
    t->w = w; t->h = h; t->rgba = (float *)malloc(sizeof(float) * 4 * (size_t)w * h);
    size_t n = (size_t)w * h;
    const uint8_t *b = (const uint8_t *)bytes; const float *f = (const float *)bytes; const uint16_t *hf = (const uint16_t *)bytes;
    for (size_t i = 0; i < n; i++) {
        float *o = &t->rgba[4 * i]; o[0] = o[1] = o[2] = 0.0f; o[3] = 1.0f;
        switch (fmt) {
            case MSNE_FORMAT_R8G8B8A8_SRGB: o[0] = c->srgb_lut[b[4 * i]]; o[1] = c->srgb_lut[b[4 * i + 1]]; o[2] = c->srgb_lut[b[4 * i + 2]]; o[3] = (float)b[4 * i + 3] / 255.0f; break;
            case MSNE_FORMAT_R8G8_UNORM: o[0] = (float)b[2 * i] / 255.0f; o[1] = (float)b[2 * i + 1] / 255.0f; break;
            case MSNE_FORMAT_R8_UNORM: o[0] = (float)b[i] / 255.0f; break;
            case MSNE_FORMAT_R32G32B32A32_SFLOAT: o[0] = f[4 * i]; o[1] = f[4 * i + 1]; o[2] = f[4 * i + 2]; o[3] = f[4 * i + 3]; break;
            case MSNE_FORMAT_R32G32_SFLOAT: o[0] = f[2 * i]; o[1] = f[2 * i + 1]; break;
            case MSNE_FORMAT_R32_SFLOAT: o[0] = f[i]; break;
            case MSNE_FORMAT_R16G16B16A16_SFLOAT: o[0] = half_to_float(hf[4 * i]); o[1] = half_to_float(hf[4 * i + 1]); o[2] = half_to_float(hf[4 * i + 2]); o[3] = half_to_float(hf[4 * i + 3]); break;
            default: break;
        }
    }
    return c->texture_count++;
}

/* ---------------- env preprocessing ---------------- */
static void env_free(orc_envmap *e) {
    free(e->rgb);
    for (uint32_t l = 0; l < e->mip_count; l++) free(e->lum[l]);
    free(e->lum); memset(e, 0, sizeof *e);
}
static uint32_t floor_pow2(uint32_t v) { uint32_t p = 1; while (p * 2 <= v && p * 2 != 0) p *= 2; return p; }

int OrcSetBackground(OrcContext *c, const float *rgba, Extent2D ext) {
    if (!rgba || ext.width == 0 || ext.height == 0) return -1;
    env_free(&c->env);
    orc_envmap *e = &c->env;
    uint32_t S = floor_pow2(ext.height); if (S > 1024) S = 1024;   /* BackgroundManager.zig:132,154 */
    e->size = S;
    e->rgb = (float *)malloc(sizeof(float) * 4 * (size_t)S * S);
    orc_texture src; src.rgba = (float *)rgba; src.w = ext.width; src.h = ext.height;
    /* equirectangular_to_equal_area.hlsl:9-30 */
    for (uint32_t py = 0; py < S; py++) for (uint32_t px = 0; px < S; px++) {
        v3 color = V3(0, 0, 0);
        const uint32_t spd = 3;
        for (uint32_t i = 0; i < spd; i++) for (uint32_t j = 0; j < spd; j++) {
            v2 sub = V2((float)(1 + i) / (float)(spd + 1), (float)(1 + j) / (float)(spd + 1));
            v2 dst = V2(((float)px + sub.x) / (float)S, ((float)py + sub.y) / (float)S);
            v3 dir = square_to_equal_area_sphere(dst);
            v2 sph = cartesian_to_spherical(dir);
            v2 srcc = V2(sph.x / (2.0f * ORC_PI), sph.y / ORC_PI);
            float o[4]; tex_sample_bilinear(&src, srcc.x, srcc.y, 1, o);
            color = v3add(color, V3(o[0], o[1], o[2]));
        }
        float *d = &e->rgb[4 * ((size_t)py * S + px)];
        d[0] = color.x / 9.0f; d[1] = color.y / 9.0f; d[2] = color.z / 9.0f; d[3] = 1.0f;
    }
    uint32_t mips = 1; while ((S >> (mips - 1)) > 1) mips++;
    e->mip_count = mips;
    e->lum = (float **)malloc(sizeof(float *) * mips);
    e->lum[0] = (float *)malloc(sizeof(float) * (size_t)S * S);
    for (size_t i = 0; i < (size_t)S * S; i++) e->lum[0][i] = orc_luminance(V3(e->rgb[4 * i], e->rgb[4 * i + 1], e->rgb[4 * i + 2])); /* luminance.hlsl:7-15 */
    for (uint32_t l = 1; l < mips; l++) {                          /* fold.hlsl:6-17 */
        uint32_t d = S >> l, sdim = S >> (l - 1);
        e->lum[l] = (float *)malloc(sizeof(float) * (size_t)d * d);
        const float *sp = e->lum[l - 1];
        for (uint32_t y = 0; y < d; y++) for (uint32_t x = 0; x < d; x++)
            e->lum[l][(size_t)y * d + x] = sp[(size_t)(2 * y) * sdim + 2 * x] + sp[(size_t)(2 * y) * sdim + 2 * x + 1]
                                         + sp[(size_t)(2 * y + 1) * sdim + 2 * x] + sp[(size_t)(2 * y + 1) * sdim + 2 * x + 1];
    }
    for (uint32_t i = 0; i < c->sensor_count; i++) c->sensors[i].sample_count = 0;
    return 0;
}

/* ---------------- alias table: alias_table.zig:25-92 ---------------- */
static void alias_build(const float *w, uint32_t n, orc_alias_entry *entries /* n, data prefilled */, float *sum_out) {
    float sum = 0.0f;
    for (uint32_t i = 0; i < n; i++) sum += w[i];
    uint32_t less_head = ORC_MAX_UINT, more_head = ORC_MAX_UINT;
    for (uint32_t i = 0; i < n; i++) {
        float adj = (w[i] * (float)n) / sum;
        entries[i].select = adj;
        if (adj < 1.0f) { entries[i].alias = less_head; less_head = i; }
        else { entries[i].alias = more_head; more_head = i; }
    }
    while (less_head != ORC_MAX_UINT && more_head != ORC_MAX_UINT) {
        uint32_t less = less_head; less_head = entries[less].alias;
        uint32_t more = more_head; more_head = entries[more].alias;
        entries[less].alias = more;
        entries[more].select = (entries[more].select + entries[less].select) - 1.0f;
        if (entries[more].select < 1.0f) { entries[more].alias = less_head; less_head = more; }
        else { entries[more].alias = more_head; more_head = more; }
    }
    while (less_head != ORC_MAX_UINT) { uint32_t less = less_head; less_head = entries[less].alias; entries[less].select = 1.0f; }
    *sum_out = sum;
}
/* exposed for golden-vector tests */
void OrcBuildAliasTable(const float *weights, uint32_t n, uint32_t *alias_out, float *select_out, float *sum_out) {
    orc_alias_entry *e = (orc_alias_entry *)calloc(n ? n : 1, sizeof(orc_alias_entry));
    alias_build(weights, n, e, sum_out);
    for (uint32_t i = 0; i < n; i++) { alias_out[i] = e[i].alias; select_out[i] = e[i].select; }
    free(e);
}

/* ---------------- accel: Accel.zig:312-563 ---------------- */
static void rebuild_accel(OrcContext *c) {
    for (uint32_t i = 0; i < c->blas_count; i++) orc_bvh_free(&c->blases[i]);
    free(c->blases); free(c->blas_key_off); free(c->blas_key_len); free(c->blas_keys);
    c->blases = NULL; c->blas_key_off = c->blas_key_len = NULL; c->blas_keys = NULL; c->blas_count = 0; c->blas_keys_len = 0;
    for (uint32_t i = 0; i < c->instance_count; i++) {
        orc_instance *in = &c->instances[i];
        uint32_t *key = (uint32_t *)malloc(sizeof(uint32_t) * (in->geo_count ? in->geo_count : 1));
        for (uint32_t g = 0; g < in->geo_count; g++) key[g] = c->geometries[in->geo_offset + g].mesh;
        uint32_t found = ORC_MAX_UINT;
        for (uint32_t b = 0; b < c->blas_count && found == ORC_MAX_UINT; b++)   /* dedup by mesh list, Accel.zig:315-343 */
            if (c->blas_key_len[b] == in->geo_count && memcmp(&c->blas_keys[c->blas_key_off[b]], key, sizeof(uint32_t) * in->geo_count) == 0) found = b;
        if (found == ORC_MAX_UINT) {
            found = c->blas_count++;
            c->blases = (orc_bvh *)realloc(c->blases, sizeof(orc_bvh) * c->blas_count);
            c->blas_key_off = (uint32_t *)realloc(c->blas_key_off, sizeof(uint32_t) * c->blas_count);
            c->blas_key_len = (uint32_t *)realloc(c->blas_key_len, sizeof(uint32_t) * c->blas_count);
            c->blas_keys = (uint32_t *)realloc(c->blas_keys, sizeof(uint32_t) * (c->blas_keys_len + in->geo_count + 1));
            memcpy(&c->blas_keys[c->blas_keys_len], key, sizeof(uint32_t) * in->geo_count);
            c->blas_key_off[found] = c->blas_keys_len; c->blas_key_len[found] = in->geo_count; c->blas_keys_len += in->geo_count;
            orc_build_blas(c, &c->blases[found], key, in->geo_count);
        }
        in->blas = found;
        in->world_to_instance = m34_inverse_affine(&in->transform);                 /* Accel.zig:430-432 */
        free(key);
    }
    orc_build_tlas(c);
    /* emissive-triangle alias table, Accel.zig:491-539 */
    uint32_t n = 0;
    for (uint32_t i = 0; i < c->instance_count; i++) for (uint32_t g = 0; g < c->instances[i].geo_count; g++) {
        const orc_geometry *ge = &c->geometries[c->instances[i].geo_offset + g];
        if (ge->sampled) n += c->meshes[ge->mesh].index_count;
    }
    free(c->alias);
    c->alias = (orc_alias_entry *)calloc((size_t)n + 1, sizeof(orc_alias_entry));
    float *w = (float *)malloc(sizeof(float) * (n ? n : 1));
    uint32_t k = 0;
    for (uint32_t i = 0; i < c->instance_count; i++) for (uint32_t g = 0; g < c->instances[i].geo_count; g++) {
        const orc_geometry *ge = &c->geometries[c->instances[i].geo_offset + g];
        if (!ge->sampled) continue;
        const orc_mesh *m = &c->meshes[ge->mesh];
        for (uint32_t p = 0; p < m->index_count; p++, k++) {
            v3 p0 = m34_mul_point(&c->instances[i].transform, m->positions[m->indices[3 * p]]);
            v3 p1 = m34_mul_point(&c->instances[i].transform, m->positions[m->indices[3 * p + 1]]);
            v3 p2 = m34_mul_point(&c->instances[i].transform, m->positions[m->indices[3 * p + 2]]);
            w[k] = v3length(v3cross(v3sub(p1, p0), v3sub(p2, p0))) / 2.0f;
            c->alias[1 + k].instance = i; c->alias[1 + k].geometry = g; c->alias[1 + k].primitive = p;
        }
    }
    float sum = 0.0f;
    alias_build(w, n, c->alias + 1, &sum);
    c->alias[0].alias = n; c->alias[0].select = sum;
    free(w);
    c->accel_dirty = 0;
}

/* ---------------- API ---------------- */
OrcContext *OrcCreate(const MsneConfig *cfg) {
    OrcContext *c = (OrcContext *)calloc(1, sizeof(OrcContext));
    c->opts.samples_per_run = 1; c->opts.max_bounces = 1024; /* hydra.zig:97-105 */
    c->tile_size = (cfg && cfg->tile_size) ? cfg->tile_size : 64;
    c->shard_index = cfg ? cfg->shard_index : 0;
    c->shard_count = (cfg && cfg->shard_count) ? cfg->shard_count : 1;
    c->threads = 1;
    for (int i = 0; i < 256; i++) {
        double v = i / 255.0;
        c->srgb_lut[i] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4));
    }
    float white[4] = { 1, 1, 1, 1 }; Extent2D one = { 1, 1 };
    OrcSetBackground(c, white, one);     /* addDefaultBackground BackgroundManager.zig:116 */
    c->alias = (orc_alias_entry *)calloc(1, sizeof(orc_alias_entry));
    c->accel_dirty = 1;
    return c;
}
void OrcDestroy(OrcContext *c) {
    if (!c) return;
    for (uint32_t i = 0; i < c->texture_count; i++) free(c->textures[i].rgba);
    free(c->textures);
    for (uint32_t i = 0; i < c->mesh_count; i++) { free(c->meshes[i].positions); free(c->meshes[i].normals); free(c->meshes[i].texcoords); free(c->meshes[i].indices); }
    free(c->meshes); free(c->materials); free(c->geometries); free(c->instances);
    for (uint32_t i = 0; i < c->blas_count; i++) orc_bvh_free(&c->blases[i]);
    free(c->blases); free(c->blas_key_off); free(c->blas_key_len); free(c->blas_keys);
    orc_bvh_free(&c->tlas); free(c->alias); env_free(&c->env);
    for (uint32_t i = 0; i < c->sensor_count; i++) free(c->sensors[i].film);
    free(c->sensors); free(c->lenses); free(c);
}
void OrcSetThreads(OrcContext *c, int n) { c->threads = n < 1 ? 1 : n; }

int64_t OrcCreateMesh(OrcContext *c, const F32x3 *positions, const F32x3 *normals, const F32x2 *texcoords, size_t position_count, size_t attribute_count, const U32x3 *indices, size_t index_count) {
    c->meshes = (orc_mesh *)realloc(c->meshes, sizeof(orc_mesh) * (c->mesh_count + 1));
    orc_mesh *m = &c->meshes[c->mesh_count]; memset(m, 0, sizeof *m);
    m->position_count = (uint32_t)position_count; m->attribute_count = (uint32_t)attribute_count; m->index_count = (uint32_t)index_count;
    m->positions = (v3 *)malloc(sizeof(v3) * (position_count ? position_count : 1)); memcpy(m->positions, positions, sizeof(v3) * position_count);
    if (normals) { m->normals = (v3 *)malloc(sizeof(v3) * attribute_count); memcpy(m->normals, normals, sizeof(v3) * attribute_count); }
    if (texcoords) { m->texcoords = (v2 *)malloc(sizeof(v2) * attribute_count); memcpy(m->texcoords, texcoords, sizeof(v2) * attribute_count); }
    m->indices = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (index_count ? index_count : 1)); memcpy(m->indices, indices, sizeof(uint32_t) * 3 * index_count);
    return c->mesh_count++;
}
int64_t OrcCreateTexture(OrcContext *c, const void *bytes, Extent2D e, int fmt) { return add_texture(c, bytes, e.width, e.height, fmt); }
int64_t OrcCreateSolidTexture1(OrcContext *c, float v) { return add_texture(c, &v, 1, 1, MSNE_FORMAT_R32_SFLOAT); }
int64_t OrcCreateSolidTexture2(OrcContext *c, F32x2 v) { return add_texture(c, &v, 1, 1, MSNE_FORMAT_R32G32_SFLOAT); }
int64_t OrcCreateSolidTexture3(OrcContext *c, F32x3 v) { float f[4] = { v.x, v.y, v.z, 0.0f }; return add_texture(c, f, 1, 1, MSNE_FORMAT_R32G32B32A32_SFLOAT); }
int64_t OrcCreateMaterial(OrcContext *c, const MsneMaterialDesc *d) {
    c->materials = (orc_material *)realloc(c->materials, sizeof(orc_material) * (c->material_count + 1));
    orc_material *m = &c->materials[c->material_count];
    m->normal = d->normal; m->emissive = d->emissive; m->type = d->type; m->color = d->color; m->metalness = d->metalness; m->roughness = d->roughness; m->ior = d->ior;
    return c->material_count++;
}
int OrcSetMaterial(OrcContext *c, uint32_t h, const MsneMaterialDesc *d) {
    if (h >= c->material_count) return -1;
    orc_material *m = &c->materials[h];
    m->normal = d->normal; m->emissive = d->emissive; m->type = d->type; m->color = d->color; m->metalness = d->metalness; m->roughness = d->roughness; m->ior = d->ior;
    return 0;
}
static void clear_all_sensors(OrcContext *c) { for (uint32_t i = 0; i < c->sensor_count; i++) c->sensors[i].sample_count = 0; }
int64_t OrcCreateInstance(OrcContext *c, Mat3x4 t, const Geometry *geos, size_t n, bool visible) {
    c->instances = (orc_instance *)realloc(c->instances, sizeof(orc_instance) * (c->instance_count + 1));
    orc_instance *in = &c->instances[c->instance_count]; memset(in, 0, sizeof *in);
    memcpy(&in->transform, &t, sizeof(m34));
    in->visible = visible; in->geo_offset = c->geometry_count; in->geo_count = (uint32_t)n;
    c->geometries = (orc_geometry *)realloc(c->geometries, sizeof(orc_geometry) * (c->geometry_count + n + 1));
    for (size_t g = 0; g < n; g++) { orc_geometry *o = &c->geometries[c->geometry_count + g]; o->mesh = geos[g].mesh; o->material = geos[g].material; o->sampled = geos[g].sampled ? 1u : 0u; }
    c->geometry_count += (uint32_t)n;
    c->accel_dirty = 1; clear_all_sensors(c);
    return c->instance_count++;
}
void OrcSetInstanceTransform(OrcContext *c, uint32_t h, Mat3x4 t) { memcpy(&c->instances[h].transform, &t, sizeof(m34)); c->accel_dirty = 1; clear_all_sensors(c); }
void OrcSetInstanceVisibility(OrcContext *c, uint32_t h, bool v) { c->instances[h].visible = v; c->accel_dirty = 1; clear_all_sensors(c); }
/* Accel.recordUpdateSingleMaterial (Accel.zig:609-628): one field of the flat geometry table (index = the instance's custom index + geometry_index, online/main.zig:225);
 * the acceleration structure and the alias table do not depend on it; sensors are left alone (the caller clears: online/main.zig:231) */
int OrcSetGeometryMaterial(OrcContext *c, uint32_t h, uint32_t geometry_index, uint32_t material) {
    if (h >= c->instance_count || geometry_index >= c->instances[h].geo_count) return -1;
    if (material >= c->material_count) return -2;
    c->geometries[c->instances[h].geo_offset + geometry_index].material = material;
    return 0;
}
/* test infrastructure: the search with its box culls switched off (orc_bvh.c scene_exhaustive); 1 = every instance entered, 2 = and every triangle tested */
void OrcSetExhaustiveSearch(OrcContext *c, int level) { c->exhaustive = level; }
int OrcSetPipeline(OrcContext *c, const MsnePipelineOpts *o) { c->opts = *o; clear_all_sensors(c); return 0; }
int64_t OrcCreateSensor(OrcContext *c, Extent2D e) {
    c->sensors = (orc_sensor *)realloc(c->sensors, sizeof(orc_sensor) * (c->sensor_count + 1));
    orc_sensor *s = &c->sensors[c->sensor_count];
    s->w = e.width; s->h = e.height; s->sample_count = 0; s->film = (float *)calloc((size_t)e.width * e.height * 4, sizeof(float));
    return c->sensor_count++;
}
float *OrcGetSensorData(OrcContext *c, uint32_t s) { return c->sensors[s].film; }
uint32_t OrcGetSampleCount(OrcContext *c, uint32_t s) { return c->sensors[s].sample_count; }
void OrcClearSensor(OrcContext *c, uint32_t s) { c->sensors[s].sample_count = 0; }
int64_t OrcCreateLens(OrcContext *c, Lens l) {
    c->lenses = (Lens *)realloc(c->lenses, sizeof(Lens) * (c->lens_count + 1));
    c->lenses[c->lens_count] = l; return c->lens_count++;
}
void OrcSetLens(OrcContext *c, uint32_t h, Lens l) { c->lenses[h] = l; clear_all_sensors(c); }

typedef struct { OrcContext *c; orc_sensor *s; const Lens *lens; volatile uint32_t *next; uint32_t ntiles, tiles_x; orc_counters cnt; } worker_arg;

static void *worker(void *p) {
    worker_arg *a = (worker_arg *)p;
    const uint32_t ts = a->c->tile_size;
    /* work unit = a band of 4 rows of a tile (many more units than threads: the host may have 256 of them); counters are kept on the
     * worker's own stack — neighbouring worker_arg records share cache lines, and a counter is bumped for every node visit */
    const uint32_t band = 4, bands = (ts + band - 1) / band;
    orc_counters cnt; memset(&cnt, 0, sizeof cnt);
    for (;;) {
        uint32_t u = __atomic_fetch_add(a->next, 1, __ATOMIC_RELAXED);
        uint32_t t = u / bands, b = u % bands;
        if (t >= a->ntiles) break;
        if (t % a->c->shard_count != a->c->shard_index) continue;          /* tile t -> shard t mod G */
        uint32_t tx = t % a->tiles_x, ty = t / a->tiles_x;
        for (uint32_t y = ty * ts + b * band; y < ty * ts + (b + 1) * band && y < (ty + 1) * ts && y < a->s->h; y++)
            for (uint32_t x = tx * ts; x < (tx + 1) * ts && x < a->s->w; x++)
                render_pixel(a->c, a->s, a->lens, x, y, &cnt);
    }
    a->cnt = cnt;
    return NULL;
}

/* `launches` x HdMoonshineRender (hydra.zig:145-363) */
int OrcRender(OrcContext *c, uint32_t sensor, uint32_t lens, uint32_t launches) {
    if (sensor >= c->sensor_count || lens >= c->lens_count) return -1;
    if (c->accel_dirty) rebuild_accel(c);
    orc_sensor *s = &c->sensors[sensor];
    const uint32_t ts = c->tile_size;
    uint32_t tiles_x = (s->w + ts - 1) / ts, tiles_y = (s->h + ts - 1) / ts;
    for (uint32_t l = 0; l < launches; l++) {
        volatile uint32_t next = 0;
        int nt = c->threads; if (nt > 256) nt = 256;
        pthread_t th[256]; worker_arg args[256];
        for (int i = 0; i < nt; i++) {
            memset(&args[i], 0, sizeof args[i]);
            args[i].c = c; args[i].s = s; args[i].lens = &c->lenses[lens]; args[i].next = &next; args[i].ntiles = tiles_x * tiles_y; args[i].tiles_x = tiles_x;
            if (nt > 1) pthread_create(&th[i], NULL, worker, &args[i]);
        }
        if (nt == 1) worker(&args[0]);
        for (int i = 0; i < nt; i++) {
            if (nt > 1) pthread_join(th[i], NULL);
            c->counters.closest_rays += args[i].cnt.closest_rays; c->counters.shadow_rays += args[i].cnt.shadow_rays;
            c->counters.samples += args[i].cnt.samples; c->counters.surface_hits += args[i].cnt.surface_hits;
            c->counters.node_visits += args[i].cnt.node_visits; c->counters.tri_tests += args[i].cnt.tri_tests;
            c->counters.shadow_node_visits += args[i].cnt.shadow_node_visits; c->counters.shadow_tri_tests += args[i].cnt.shadow_tri_tests;
        }
        s->sample_count += c->opts.samples_per_run;
    }
    return 0;
}
/* trace one camera path (sample index k, pixel x,y): returns radiance in rgb[3], hit records in rec (12 floats each: instance, primitive, t, u, v, ray direction, ray origin, geometry) */
uint32_t OrcDebugPath(OrcContext *c, uint32_t sensor, uint32_t lens, uint32_t k, uint32_t x, uint32_t y, float rgb[3], float *rec, uint32_t cap, uint64_t counts[2] /* closest, shadow rays of the path; may be NULL */) {
    if (c->accel_dirty) rebuild_accel(c);
    orc_sensor *s = &c->sensors[sensor]; orc_counters cnt; memset(&cnt, 0, sizeof cnt);
    orc_rng rng = rng_from_seed(k, x, y);
    v2 r1; r1.x = rng_get_float(&rng); r1.y = rng_get_float(&rng);
    v2 g = square_to_gaussian(r1);
    v2 center = V2(0.5f + 0.5f * g.x, 0.5f + 0.5f * g.y);
    v2 uv = V2(((float)x + center.x) / (float)s->w, ((float)y + center.y) / (float)s->h);
    if (c->opts.flip_image) uv.y = 1.0f - uv.y;
    v2 r2; r2.x = rng_get_float(&rng); r2.y = rng_get_float(&rng);
    v3 O, D; generate_ray(&c->lenses[lens], s->w, s->h, uv, r2, &O, &D);
    g_dbg = rec; g_dbg_n = 0; g_dbg_cap = cap;
    v3 L = incoming_radiance(c, O, D, &rng, &cnt);
    g_dbg = NULL;
    rgb[0] = L.x; rgb[1] = L.y; rgb[2] = L.z;
    if (counts) { counts[0] = cnt.closest_rays; counts[1] = cnt.shadow_rays; }
    return g_dbg_n;
}
void OrcGetCounters(OrcContext *c, uint64_t out[8]) {
    out[6] = c->counters.shadow_node_visits; out[7] = c->counters.shadow_tri_tests;
    out[0] = c->counters.closest_rays; out[1] = c->counters.shadow_rays; out[2] = c->counters.samples;
    out[3] = c->counters.surface_hits; out[4] = c->counters.node_visits; out[5] = c->counters.tri_tests;
}
void OrcResetCounters(OrcContext *c) { memset(&c->counters, 0, sizeof c->counters); }

/* direct probes for parity tests of single functions */
int OrcTraceClosest(OrcContext *c, const float o[3], const float d[3], float tmax, uint32_t out_ids[3], float out_tuv[3]) {
    if (c->accel_dirty) rebuild_accel(c);
    orc_hit h; orc_counters cnt; memset(&cnt, 0, sizeof cnt);
    int r = orc_closest_hit(c, V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmax, &h, &cnt);
    out_ids[0] = h.inst; out_ids[1] = h.geo; out_ids[2] = h.prim; out_tuv[0] = h.t; out_tuv[1] = h.u; out_tuv[2] = h.v;
    return r;
}
int OrcTraceShadow(OrcContext *c, const float o[3], const float d[3], float tmax) {
    if (c->accel_dirty) rebuild_accel(c);
    orc_counters cnt; memset(&cnt, 0, sizeof cnt);
    return orc_shadow_hit(c, V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), tmax, &cnt);
}
void OrcGenerateRay(const Lens *lens, uint32_t W, uint32_t H, float u, float v, float r0, float r1, float out[6]) {
    v3 O, D; generate_ray(lens, W, H, V2(u, v), V2(r0, r1), &O, &D);
    out[0] = O.x; out[1] = O.y; out[2] = O.z; out[3] = D.x; out[4] = D.y; out[5] = D.z;
}
uint32_t OrcEnvSize(OrcContext *c) { return c->env.size; }
const float *OrcEnvRgb(OrcContext *c) { return c->env.rgb; }
const float *OrcEnvLum(OrcContext *c, uint32_t level) { return level < c->env.mip_count ? c->env.lum[level] : NULL; }
uint32_t OrcAliasCount(OrcContext *c) { if (c->accel_dirty) rebuild_accel(c); return c->alias[0].alias + 1; }
const void *OrcAliasTable(OrcContext *c) { if (c->accel_dirty) rebuild_accel(c); return c->alias; }
void OrcWorldToInstance(OrcContext *c, uint32_t inst, float out[12]) { if (c->accel_dirty) rebuild_accel(c); memcpy(out, &c->instances[inst].world_to_instance, 48); }

/* scalar function probes (golden vectors; also used to check device math bit-for-bit) */
void OrcMathProbe(int fn, const float *in, float *out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = det_sinf(in[i]); break;
            case 1: out[i] = det_cosf(in[i]); break;
            case 2: out[i] = det_logf(in[i]); break;
            case 3: out[i] = det_acosf(in[i]); break;
            case 4: out[i] = det_tanf(in[i]); break;
            case 5: out[i] = det_atan2f(in[2 * i], in[2 * i + 1]); break;
            default: out[i] = 0.0f;
        }
    }
}
void OrcSquareToEqualAreaSphere(const float *uv, float *out, uint32_t n) { for (uint32_t i = 0; i < n; i++) { v3 d = square_to_equal_area_sphere(V2(uv[2 * i], uv[2 * i + 1])); out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z; } }
void OrcSquareToEqualAreaSphereInverse(const float *d, float *out, uint32_t n) { for (uint32_t i = 0; i < n; i++) { v2 u = square_to_equal_area_sphere_inverse(V3(d[3 * i], d[3 * i + 1], d[3 * i + 2])); out[2 * i] = u.x; out[2 * i + 1] = u.y; } }
void OrcOffsetAlongNormal(const float *p, const float *n, float *out, uint32_t cnt) { for (uint32_t i = 0; i < cnt; i++) { v3 r = offset_along_normal(V3(p[3 * i], p[3 * i + 1], p[3 * i + 2]), V3(n[3 * i], n[3 * i + 1], n[3 * i + 2])); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; } }
void OrcRngFloats(uint32_t s, uint32_t x, uint32_t y, float *out, uint32_t n, uint32_t *state0) { orc_rng r = rng_from_seed(s, x, y); *state0 = r.state; for (uint32_t i = 0; i < n; i++) out[i] = rng_get_float(&r); }
uint32_t OrcPcg(uint32_t a) { return hash_pcg(a); }
/* BSDF probes: type, params {color rgb, metalness, roughness, ior}, wi, wo, sq -> pdf, eval rgb, sample dir + pdf */
void OrcBsdfProbe(uint32_t type, const float params[6], const float wi[3], const float wo[3], const float sq[2], float out[8]) {
    orc_mat m; m.type = type; m.color = V3(params[0], params[1], params[2]); m.metalness = params[3];
    m.alpha = orc_maxf(params[4] * params[4], 0.001f); m.ior = params[5];
    v3 a = V3(wi[0], wi[1], wi[2]), b = V3(wo[0], wo[1], wo[2]);
    out[0] = material_pdf(&m, a, b);
    v3 e = material_eval(&m, a, b); out[1] = e.x; out[2] = e.y; out[3] = e.z;
    orc_msample s = material_sample(&m, b, V2(sq[0], sq[1]));
    out[4] = s.dirFs.x; out[5] = s.dirFs.y; out[6] = s.dirFs.z; out[7] = s.pdf;
}

/* Batch probes of the shading functions: `n` records of PROBE_IN[fn] floats in, PROBE_OUT[fn] floats out (the table is
 * mirrored by MsneShadeProbe in the product and by tests/second_source.py, the independent float64 restatement).
 *  0 bsdf        {type, color rgb, metalness, roughness, ior, wi xyz, wo xyz, sq xy} -> {pdf, eval rgb, sample dir xyz, sample pdf}
 *  1 env sample  {rand xy} -> {dir xyz, radiance rgb, pdf}            (context environment; unoccluded)
 *  2 env eval    {dir xyz} -> {radiance rgb, pdf}
 *  3 env incomingRadiance {dir xyz} -> {rgb}
 *  4 squareToEqualAreaSphere {uv} -> {dir}          5 ...Inverse {dir} -> {uv}
 *  6 squareToTriangle {sq} -> {ab}                  7 squareToGaussian {sq} -> {xy}
 *  8 squareToCosineHemisphere {sq} -> {dir}         9 Fresnel::dielectric {cos, etaI, etaT} -> {F}
 * 10 offsetAlongNormal {p, n} -> {p'}              11 coordinateSystem {v1} -> {v2, v3}
 * 12 areaMeasureToSolidAngleMeasure {pos1, pos2, dir1, dir2} -> {x}
 * 13 GGX {alpha, a xyz, b xyz} -> {D(a), Lambda(a), G(a, b)}
 * 14 refractDir {wi, n, eta} -> {dir}              15 powerHeuristic {numf, fPdf, numg, gPdf} -> {w}
 * 16 Frame: {n xyz, s xyz, v xyz} -> reorthogonalize(n, s) then {worldToFrame(v), frameToWorld(v)}
 * 17 dTextures[i].SampleLevel(dTextureSampler, uv, 0) {texture index, u, v} -> {rgba}   (context textures; linear, repeat)
 * 18 MeshAttributes::lookupAndInterpolate + inWorld {p0 p1 p2, t0 t1 t2, n0 n1 n2, attribs, flags (1 texcoords | 2 normals), toWorld 3x4, toMesh 3x4}
 *    -> {position, texcoord, triangleFrame n s t, frame n s t}
 * 19 getTextureFrame {normal texel rgb, tangent frame n s t, two_component} -> {frame n s t}
 * 20 Camera::generateRay {origin, forward, up, vfov, aperture, focus_distance, width, height, uv, rand} -> {origin, direction} */
static const uint32_t PROBE_IN[21]  = { 15, 2, 3, 3, 2, 3, 2, 2, 2, 3, 6, 3, 12, 7, 7, 4, 9, 3, 51, 13, 18 };
static const uint32_t PROBE_OUT[21] = {  8, 7, 4, 3, 3, 2, 2, 2, 3, 1, 3, 6,  1, 3, 3, 1, 6, 4, 23,  9,  6 };
int OrcProbeBatch(OrcContext *c, int fn, const float *in, uint32_t n, float *out) {
    if (fn < 0 || fn > 20) return -1;
    orc_counters cnt; memset(&cnt, 0, sizeof cnt);
    for (uint32_t i = 0; i < n; i++) {
        const float *a = in + (size_t)i * PROBE_IN[fn]; float *o = out + (size_t)i * PROBE_OUT[fn];
        switch (fn) {
            case 0: OrcBsdfProbe((uint32_t)a[0], a + 1, a + 7, a + 10, a + 13, o); break;
            case 1: { orc_lsample s = env_sample(c, V3(0, 0, 0), V3(0, 0, 1), V2(a[0], a[1]), &cnt);
                      o[0] = s.dirWs.x; o[1] = s.dirWs.y; o[2] = s.dirWs.z; o[3] = s.radiance.x; o[4] = s.radiance.y; o[5] = s.radiance.z; o[6] = s.pdf; break; }
            case 2: { v3 r; float p; env_eval(c, V3(a[0], a[1], a[2]), &r, &p); o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = p; break; }
            case 3: { v3 r = env_incoming_radiance(c, V3(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
            case 4: { v3 d = square_to_equal_area_sphere(V2(a[0], a[1])); o[0] = d.x; o[1] = d.y; o[2] = d.z; break; }
            case 5: { v2 u = square_to_equal_area_sphere_inverse(V3(a[0], a[1], a[2])); o[0] = u.x; o[1] = u.y; break; }
            case 6: { v2 u = square_to_triangle(V2(a[0], a[1])); o[0] = u.x; o[1] = u.y; break; }
            case 7: { v2 u = square_to_gaussian(V2(a[0], a[1])); o[0] = u.x; o[1] = u.y; break; }
            case 8: { v3 d = square_to_cosine_hemisphere(V2(a[0], a[1])); o[0] = d.x; o[1] = d.y; o[2] = d.z; break; }
            case 9: o[0] = fresnel_dielectric(a[0], a[1], a[2]); break;
            case 10: { v3 r = offset_along_normal(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5])); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
            case 11: { v3 p, q; coordinate_system(V3(a[0], a[1], a[2]), &p, &q); o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; break; }
            case 12: o[0] = area_to_solid_angle(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), V3(a[6], a[7], a[8]), V3(a[9], a[10], a[11])); break;
            case 13: o[0] = ggx_D(a[0], V3(a[1], a[2], a[3])); o[1] = ggx_Lambda(a[0], V3(a[1], a[2], a[3])); o[2] = ggx_G(a[0], V3(a[1], a[2], a[3]), V3(a[4], a[5], a[6])); break;
            case 14: { v3 r = refract_dir(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), a[6]); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
            case 15: o[0] = power_heuristic((uint32_t)a[0], a[1], (uint32_t)a[2], a[3]); break;
            case 16: { orc_frame f; f.n = V3(a[0], a[1], a[2]); f.s = V3(a[3], a[4], a[5]); f.t = V3(0, 0, 0); frame_reorthogonalize(&f);
                       v3 p = frame_world_to_frame(&f, V3(a[6], a[7], a[8])), q = frame_frame_to_world(&f, V3(a[6], a[7], a[8]));
                       o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; break; }
            case 17: { uint32_t t = (uint32_t)a[0]; if (!c || t >= c->texture_count) return -1; tex_sample_bilinear(&c->textures[t], a[1], a[2], 0, o); break; }
            case 18: {
                uint32_t flags = (uint32_t)a[26];
                int ht = (flags & 1u) != 0, hn = (flags & 2u) != 0;
                v2 t0 = ht ? V2(a[9], a[10]) : V2(0, 0), t1 = ht ? V2(a[11], a[12]) : V2(1, 0), t2 = ht ? V2(a[13], a[14]) : V2(1, 1);
                m34 tw, tm;
                for (int r = 0; r < 3; r++) for (int k = 0; k < 4; k++) { tw.m[r][k] = a[27 + 4 * r + k]; tm.m[r][k] = a[39 + 4 * r + k]; }
                orc_attrs at = mesh_attributes_core(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), V3(a[6], a[7], a[8]), t0, t1, t2, V3(a[15], a[16], a[17]), V3(a[18], a[19], a[20]), V3(a[21], a[22], a[23]),
                                                    hn, V3(1.0f - a[24] - a[25], a[24], a[25]), &tw, &tm);
                o[0] = at.position.x; o[1] = at.position.y; o[2] = at.position.z; o[3] = at.texcoord.x; o[4] = at.texcoord.y;
                const orc_frame *fr[2] = { &at.triangleFrame, &at.frame };
                for (int k = 0; k < 2; k++) { float *q = o + 5 + 9 * k; q[0] = fr[k]->n.x; q[1] = fr[k]->n.y; q[2] = fr[k]->n.z; q[3] = fr[k]->s.x; q[4] = fr[k]->s.y; q[5] = fr[k]->s.z; q[6] = fr[k]->t.x; q[7] = fr[k]->t.y; q[8] = fr[k]->t.z; }
                break; }
            case 19: {
                orc_frame tf; tf.n = V3(a[3], a[4], a[5]); tf.s = V3(a[6], a[7], a[8]); tf.t = V3(a[9], a[10], a[11]);
                float texel[4] = { a[0], a[1], a[2], 1.0f };
                orc_frame f = texture_frame_from_texel(texel, a[12] != 0.0f, &tf);
                o[0] = f.n.x; o[1] = f.n.y; o[2] = f.n.z; o[3] = f.s.x; o[4] = f.s.y; o[5] = f.s.z; o[6] = f.t.x; o[7] = f.t.y; o[8] = f.t.z;
                break; }
            case 20: {
                if (!(a[12] >= 1.0f && a[13] >= 1.0f)) return -1;
                Lens lens; memset(&lens, 0, sizeof lens);
                lens.origin.x = a[0]; lens.origin.y = a[1]; lens.origin.z = a[2]; lens.forward.x = a[3]; lens.forward.y = a[4]; lens.forward.z = a[5];
                lens.up.x = a[6]; lens.up.y = a[7]; lens.up.z = a[8]; lens.vfov = a[9]; lens.aperture = a[10]; lens.focus_distance = a[11];
                v3 O, D; generate_ray(&lens, (uint32_t)a[12], (uint32_t)a[13], V2(a[14], a[15]), V2(a[16], a[17]), &O, &D);
                o[0] = O.x; o[1] = O.y; o[2] = O.z; o[3] = D.x; o[4] = D.y; o[5] = D.z;
                break; }
        }
    }
    return 0;
}
