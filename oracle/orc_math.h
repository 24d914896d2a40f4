/*
 * ORACLE — test infrastructure only.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build or call anything under oracle/.
 *
 * orc_math.h: scalar f32 restatement of the reference's shader utilities:
 *   shaders/utils/random.hlsl:8-46      (PCG RNG)
 *   shaders/utils/math.hlsl:3-64        (constants, luminance, faceForward, offsetAlongNormal, coordinateSystem)
 *   shaders/utils/mappings.hlsl:5-126   (sampling warps, coinFlipRemap)
 *   shaders/hrtsystem/reflection_frame.hlsl:3-84
 *
 * Arithmetic contract: every operation is a single IEEE-754 binary32 operation
 * (compile with -ffp-contract=off, no -ffast-math).  Transcendentals (sin, cos, log,
 * atan2, acos, tan) are implemented here from +,-,*,/ and sqrt only ("det_*"), because
 * the reference's are whatever the Vulkan driver provides (unpinnable) and libm's differ
 * from the GPU's by ulps; a from-scratch polynomial version is evaluated identically
 * on the host and on the device, which is what makes GPU==oracle bit-exact.
 * Known deviations from the literal HLSL: pow(x,2) -> x*x,
 * pow(1-c,5) -> repeated multiplication.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float m[3][4]; } m34; /* row-major 3x4, like reference Mat3x4 (vector.zig:245) */

/* math.hlsl:3-7 */
#define ORC_PI 3.14159265f
#define ORC_EPSILON 0.000000119f
#define ORC_INFINITY 1000000000000.0f
#define ORC_MAX_UINT 0xFFFFFFFFu
#define ORC_AIR_IOR 1.000277f

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline v2 V2(float x, float y) { v2 r = { x, y }; return r; }
static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3div(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 v3neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3length(v3 a) { return sqrtf(v3dot(a, a)); }
/* normalize(v) = v * (1 / |v|): one IEEE reciprocal, three multiplications — the expression the product computes (csrc/msne_math.h);
 * HLSL leaves normalize's precision unspecified */
static inline v3 v3normalize(v3 a) { const float r = 1.0f / v3length(a); return V3(a.x * r, a.y * r, a.z * r); }
static inline float orc_minf(float a, float b) { return a < b ? a : b; }
static inline float orc_maxf(float a, float b) { return a > b ? a : b; }
static inline float orc_clampf(float x, float lo, float hi) { return orc_minf(orc_maxf(x, lo), hi); }
/* HLSL lerp(x,y,s) -> GLSL.std.450 FMix = x*(1-s) + y*s */
static inline float orc_lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline float orc_signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

/* ---------------- deterministic transcendentals ---------------- */

/* sin and cos of x, |x| < 2^13; 3-term Cody-Waite reduction by pi/4 octants and degree-7/8
 * minimax polynomials on [-pi/4, pi/4] (classic single-precision coefficients). */
static inline void det_sincosf(float x, float *s, float *c) {
    float ax = fabsf(x);
    int j = (int)(ax * 1.27323954473516f); /* 4/pi */
    j = (j + 1) & ~1;                      /* round up to even octant */
    float y = (float)j;
    float z = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float zz = z * z;
    float ps = ((-1.9515295891e-4f * zz + 8.3321608736e-3f) * zz - 1.6666654611e-1f) * zz * z + z;
    float pc = ((2.443315711809948e-5f * zz - 1.388731625493765e-3f) * zz + 4.166664568298827e-2f) * zz * zz - 0.5f * zz + 1.0f;
    int q = (j >> 1) & 3; /* quadrant of the reduced argument */
    float sv, cv;
    switch (q) {
        case 0: sv = ps; cv = pc; break;
        case 1: sv = pc; cv = -ps; break;
        case 2: sv = -ps; cv = -pc; break;
        default: sv = -pc; cv = ps; break;
    }
    *s = x < 0.0f ? -sv : sv;
    *c = cv;
}
static inline float det_sinf(float x) { float s, c; det_sincosf(x, &s, &c); return s; }
static inline float det_cosf(float x) { float s, c; det_sincosf(x, &s, &c); return c; }
static inline float det_tanf(float x) { float s, c; det_sincosf(x, &s, &c); return s / c; }

/* natural log for normal positive x (callers pass (0,1]) ; x==1 -> exactly 0 */
static inline float det_logf(float x) {
    uint32_t u = f2u(x);
    int e = (int)((u >> 23) & 0xff) - 126;             /* x = m * 2^e, m in [0.5,1) */
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                 + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
                 + 3.3333331174e-1f) * m * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = m + y;
    z = z + 0.693359375f * fe;
    return z;
}

/* atan for t >= 0 */
static inline float det_atanf_pos(float t) {
    float y0;
    if (t > 2.414213562373095f) { y0 = 1.5707963267948966f; t = -(1.0f / t); }
    else if (t > 0.4142135623730950f) { y0 = 0.7853981633974483f; t = (t - 1.0f) / (t + 1.0f); }
    else { y0 = 0.0f; }
    float z = t * t;
    float y = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t;
    return y0 + y;
}
/* full-quadrant atan2; atan2(0,0) = 0 */
static inline float det_atan2f(float y, float x) {
    float ay = fabsf(y), ax = fabsf(x);
    float a;
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    if (ax == 0.0f) a = 1.5707963267948966f;
    else a = det_atanf_pos(ay / ax);
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return y < 0.0f ? -a : a;
}
/* asin for |x| <= 0.5 */
static inline float det_asinf_small(float x) {
    float z = x * x;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
            + 1.6666752422e-1f) * z * x + x;
}
static inline float det_acosf(float x) {
    if (x > 1.0f) x = 1.0f;
    if (x < -1.0f) x = -1.0f;
    if (x > 0.5f) return 2.0f * det_asinf_small(sqrtf(0.5f * (1.0f - x)));
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * det_asinf_small(sqrtf(0.5f * (1.0f + x)));
    return 1.5707963267948966f - det_asinf_small(x);
}

/* ---------------- random.hlsl ---------------- */
static inline uint32_t hash_lcg(uint32_t a) { return a * 747796405u + 2891336453u; }          /* random.hlsl:8-12 */
static inline uint32_t hash_rxs_m_xs(uint32_t a) {                                            /* random.hlsl:15-18 */
    uint32_t b = ((a >> ((a >> 28u) + 4u)) ^ a) * 277803737u;
    return (b >> 22u) ^ b;
}
static inline uint32_t hash_pcg(uint32_t a) { return hash_rxs_m_xs(hash_lcg(a)); }            /* random.hlsl:20-22 */
typedef struct { uint32_t state; } orc_rng;
static inline orc_rng rng_from_seed(uint32_t sx, uint32_t sy, uint32_t sz) {                   /* random.hlsl:28-32 */
    orc_rng r; r.state = hash_pcg(sx + hash_pcg(sy + hash_pcg(sz))); return r;
}
static inline float rng_get_float(orc_rng *r) {                                               /* random.hlsl:38-46 */
    r->state = hash_lcg(r->state);
    uint32_t h = hash_rxs_m_xs(r->state);
    return (float)(h >> 8) * 0x1p-24f;
}

/* ---------------- math.hlsl ---------------- */
static inline float orc_luminance(v3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; } /* math.hlsl:17-21 */
static inline v3 face_forward(v3 n, v3 d) { return v3dot(n, d) > 0.0f ? n : v3neg(n); }           /* math.hlsl:23-25 */

static inline float offset_component(float p, float n) {                                        /* math.hlsl:32-42 */
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int32_t of_i = (int32_t)(n * int_scale);
    int32_t pi = (int32_t)f2u(p);
    float p_i = u2f((uint32_t)(pi + (p < 0.0f ? -of_i : of_i)));
    return fabsf(p) < origin ? p + n * float_scale : p_i;
}
static inline v3 offset_along_normal(v3 p, v3 n) {
    return V3(offset_component(p.x, n.x), offset_component(p.y, n.y), offset_component(p.z, n.z));
}
static inline void coordinate_system(v3 v1, v3 *v2o, v3 *v3o) {                                  /* math.hlsl:56-64 */
    if (fabsf(v1.x) > fabsf(v1.y)) *v2o = v3div(V3(-v1.z, 0.0f, v1.x), sqrtf(v1.x * v1.x + v1.z * v1.z));
    else *v2o = v3div(V3(0.0f, v1.z, -v1.y), sqrtf(v1.y * v1.y + v1.z * v1.z));
    *v3o = v3cross(*v2o, v1);
}

/* ---------------- mappings.hlsl ---------------- */
static inline v2 square_to_triangle(v2 sq) {                                                    /* mappings.hlsl:5-9 */
    float a = 1.0f - sqrtf(1.0f - sq.x);
    float b = sq.y * sqrtf(1.0f - sq.x);
    return V2(a, b);
}
static inline v2 square_to_gaussian(v2 sq) {                                                    /* mappings.hlsl:11-17 */
    const float u1 = 1.0f - sq.x, u2 = sq.y;
    const float r = sqrtf(-2.0f * det_logf(u1));
    const float theta = 2.0f * ORC_PI * u2;
    float s, c; det_sincosf(theta, &s, &c);
    return V2(r * c, r * s);
}
static inline v2 square_to_uniform_disk_concentric(v2 sq) {                                     /* mappings.hlsl:19-37 */
    v2 o = V2(2.0f * sq.x - 1.0f, 2.0f * sq.y - 1.0f);
    if (o.x == 0.0f && o.y == 0.0f) return V2(0.0f, 0.0f);
    float theta, r;
    if (fabsf(o.x) > fabsf(o.y)) { r = o.x; theta = (ORC_PI / 4.0f) * (o.y / o.x); }
    else { r = o.y; theta = (ORC_PI / 2.0f) - (ORC_PI / 4.0f) * (o.x / o.y); }
    float s, c; det_sincosf(theta, &s, &c);
    return V2(r * c, r * s);
}
static inline v3 square_to_cosine_hemisphere(v2 sq) {                                           /* mappings.hlsl:39-44 */
    v2 d = square_to_uniform_disk_concentric(sq);
    float z = sqrtf(orc_maxf(0.0f, 1.0f - (d.x * d.x + d.y * d.y)));
    return V3(d.x, d.y, z);
}
static inline v3 spherical_to_cartesian(float sinTheta, float cosTheta, float phi) {            /* mappings.hlsl:53-55 */
    float s, c; det_sincosf(phi, &s, &c);
    return V3(sinTheta * c, sinTheta * s, cosTheta);
}
static inline v2 cartesian_to_spherical(v3 v) {                                                 /* mappings.hlsl:59-64 */
    float p = det_atan2f(v.y, v.x);
    float phi = (p < 0.0f) ? (p + 2.0f * ORC_PI) : p;
    float theta = det_acosf(v.z);
    return V2(phi, theta);
}
static inline v3 square_to_equal_area_sphere(v2 sq) {                                           /* mappings.hlsl:67-83 */
    const v2 uv = V2(2.0f * sq.x - 1.0f, 2.0f * sq.y - 1.0f);
    const v2 uvp = V2(fabsf(uv.x), fabsf(uv.y));
    const float signedDistance = 1.0f - (uvp.x + uvp.y);
    const float d = fabsf(signedDistance);
    const float r = 1.0f - d;
    const float phi = (r == 0.0f ? 1.0f : (uvp.y - uvp.x) / r + 1.0f) * ORC_PI / 4.0f;
    float s, c; det_sincosf(phi, &s, &c);
    const float q = sqrtf(2.0f - r * r);
    return V3(orc_signf(uv.x) * (c * r * q), orc_signf(uv.y) * (s * r * q), orc_signf(signedDistance) * (1.0f - r * r));
}
static inline v2 square_to_equal_area_sphere_inverse(v3 dir) {                                  /* mappings.hlsl:85-99 */
    const v3 a = V3(fabsf(dir.x), fabsf(dir.y), fabsf(dir.z));
    const float r = sqrtf(1.0f - a.z);
    float phi = (a.x == 0.0f && a.y == 0.0f) ? 0.0f : det_atan2f(orc_minf(a.x, a.y), orc_maxf(a.x, a.y)) * 2.0f / ORC_PI;
    if (a.x < a.y) phi = 1.0f - phi;
    v2 uv = V2(r - phi * r, phi * r);
    if (dir.z < 0.0f) uv = V2(1.0f - uv.y, 1.0f - uv.x);
    uv.x *= orc_signf(dir.x); uv.y *= orc_signf(dir.y);
    return V2((uv.x + 1.0f) / 2.0f, (uv.y + 1.0f) / 2.0f);
}
static inline int coin_flip_remap(float p, float *rand) {                                       /* mappings.hlsl:103-111 */
    if (*rand < p) { *rand /= p; return 1; }
    *rand = (*rand - p) / (1.0f - p);
    return 0;
}

/* ---------------- reflection_frame.hlsl ---------------- */
typedef struct { v3 n, s, t; } orc_frame;
static inline void frame_reorthogonalize(orc_frame *f) {                                        /* reflection_frame.hlsl:32-36 */
    f->s = v3normalize(v3sub(f->s, v3scale(f->n, v3dot(f->n, f->s))));
    f->t = v3normalize(v3cross(f->n, f->s));
}
static inline v3 frame_world_to_frame(const orc_frame *f, v3 v) {                               /* :38-41 */
    return V3(v3dot(f->s, v), v3dot(f->t, v), v3dot(f->n, v));
}
static inline v3 frame_frame_to_world(const orc_frame *f, v3 v) {                               /* :43-46  mul(transpose({s,t,n}), v) */
    return V3(f->s.x * v.x + f->t.x * v.y + f->n.x * v.z,
              f->s.y * v.x + f->t.y * v.y + f->n.y * v.z,
              f->s.z * v.x + f->t.z * v.y + f->n.z * v.z);
}
static inline float frame_cos2theta(v3 v) { return v.z * v.z; }
static inline float frame_sin2theta(v3 v) { return orc_maxf(0.0f, 1.0f - frame_cos2theta(v)); }
static inline float frame_tan2theta(v3 v) { return frame_sin2theta(v) / frame_cos2theta(v); }
static inline int frame_same_hemisphere(v3 a, v3 b) { return a.z * b.z > 0.0f; }

/* matrix helpers: engine/vector.zig:277-362 */
static inline v3 m34_mul_point(const m34 *m, v3 p) { /* dot(row, (p,1)) = x*px + y*py + z*pz + w*1 */
    return V3(m->m[0][0] * p.x + m->m[0][1] * p.y + m->m[0][2] * p.z + m->m[0][3] * 1.0f,
              m->m[1][0] * p.x + m->m[1][1] * p.y + m->m[1][2] * p.z + m->m[1][3] * 1.0f,
              m->m[2][0] * p.x + m->m[2][1] * p.y + m->m[2][2] * p.z + m->m[2][3] * 1.0f);
}
static inline v3 m34_mul_vec(const m34 *m, v3 p) {
    return V3(m->m[0][0] * p.x + m->m[0][1] * p.y + m->m[0][2] * p.z,
              m->m[1][0] * p.x + m->m[1][1] * p.y + m->m[1][2] * p.z,
              m->m[2][0] * p.x + m->m[2][1] * p.y + m->m[2][2] * p.z);
}
/* mul(transpose(toMesh) as float4x3, v).xyz : out_j = sum_i v_i * M[i][j], j<3 (reflection_frame.hlsl:24-30) */
static inline v3 m34_mul_transposed(const m34 *m, v3 v) {
    return V3(m->m[0][0] * v.x + m->m[1][0] * v.y + m->m[2][0] * v.z,
              m->m[0][1] * v.x + m->m[1][1] * v.y + m->m[2][1] * v.z,
              m->m[0][2] * v.x + m->m[1][2] * v.y + m->m[2][2] * v.z);
}
/* vector.zig:350-362 + Mat3.inverse :512-520 */
static inline m34 m34_inverse_affine(const m34 *s) {
    /* p = transpose of upper 3x3 : p.x = column 0 of s etc. */
    v3 px = V3(s->m[0][0], s->m[1][0], s->m[2][0]);
    v3 py = V3(s->m[0][1], s->m[1][1], s->m[2][1]);
    v3 pz = V3(s->m[0][2], s->m[1][2], s->m[2][2]);
    v3 v = V3(s->m[0][3], s->m[1][3], s->m[2][3]);
    float det = v3dot(px, v3cross(py, pz));
    float inv = 1.0f / det;
    v3 v1 = v3scale(v3cross(py, pz), inv), v2_ = v3scale(v3cross(pz, px), inv), v3_ = v3scale(v3cross(px, py), inv);
    /* inv_p = Mat3(v1,v2,v3).transpose(): rows ix,iy,iz */
    v3 ix = V3(v1.x, v2_.x, v3_.x), iy = V3(v1.y, v2_.y, v3_.y), iz = V3(v1.z, v2_.z, v3_.z);
    /* neg_inv_p_v = (inv_p * -1).mul_vec(v) = x*v.x, then y*v.y + res, then z*v.z + res (column combination) */
    v3 nx = v3scale(ix, -1.0f), ny = v3scale(iy, -1.0f), nz = v3scale(iz, -1.0f);
    v3 res = v3scale(nx, v.x);
    res = v3add(v3scale(ny, v.y), res);
    res = v3add(v3scale(nz, v.z), res);
    m34 o;
    o.m[0][0] = ix.x; o.m[0][1] = iy.x; o.m[0][2] = iz.x; o.m[0][3] = res.x;
    o.m[1][0] = ix.y; o.m[1][1] = iy.y; o.m[1][2] = iz.y; o.m[1][3] = res.y;
    o.m[2][0] = ix.z; o.m[2][1] = iy.z; o.m[2][2] = iz.z; o.m[2][3] = res.z;
    return o;
}

#endif
