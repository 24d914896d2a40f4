// msne_math.h — f32 math shared by every kernel of the hot path (and by the host code that prepares
// kernel constants).  Restates shaders/utils/{random,math,mappings}.hlsl and
// shaders/hrtsystem/reflection_frame.hlsl of the reference (line cites on each function).
//
// Arithmetic contract: every operation is one IEEE-754 binary32 operation, in the order written —
// this translation unit set is compiled with -ffp-contract=off.  sin/cos/log/atan2/acos/tan are built
// from + - * / sqrt only (Cody–Waite reduction + minimax polynomials), so host, device and the test
// oracle evaluate them identically; the reference's own are whatever its Vulkan driver provides.
// Explicit fmaf() is used only where results do not feed radiance (box tests).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MSNE_HD __host__ __device__ __forceinline__

namespace msne {

constexpr float PI = 3.14159265f;          // math.hlsl:3
constexpr float INFINITY_F = 1000000000000.0f;  // math.hlsl:5 ("pranked": finite)
constexpr uint32_t MAX_UINT = 0xFFFFFFFFu;
constexpr float AIR_IOR = 1.000277f;       // math.hlsl:7

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct m34 { float m[3][4]; };             // row-major 3x4 (vector.zig:245)

MSNE_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
MSNE_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
MSNE_HD float absf(float x) { return u2f(f2u(x) & 0x7fffffffu); }

MSNE_HD f2 F2(float x, float y) { return f2{ x, y }; }
MSNE_HD f3 F3(float x, float y, float z) { return f3{ x, y, z }; }
MSNE_HD f3 add(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
MSNE_HD f3 sub(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
MSNE_HD f3 mul(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
MSNE_HD f3 scale(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
MSNE_HD f3 divs(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
MSNE_HD f3 neg(f3 a) { return F3(-a.x, -a.y, -a.z); }
MSNE_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MSNE_HD f3 cross(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MSNE_HD float length(f3 a) { return __builtin_sqrtf(dot(a, a)); }
// normalize(v) = v * (1 / |v|): ONE IEEE reciprocal and three multiplications (HLSL's normalize is precision-unspecified; three IEEE
// divisions per call cost k_shade ~20 % of its vector instructions — a hit evaluates about twenty normalisations).  The test oracle
// computes the same expression.
MSNE_HD f3 normalize(f3 a) { const float r = 1.0f / length(a); return F3(a.x * r, a.y * r, a.z * r); }
MSNE_HD float minf(float a, float b) { return a < b ? a : b; }
MSNE_HD float maxf(float a, float b) { return a > b ? a : b; }
MSNE_HD float clampf(float x, float lo, float hi) { return minf(maxf(x, lo), hi); }
MSNE_HD float lerpf(float a, float b, float t) { return a * (1.0f - t) + b * t; }   // HLSL lerp → FMix
MSNE_HD float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
MSNE_HD float sqrt_(float x) { return __builtin_sqrtf(x); }
MSNE_HD float floor_(float x) { return __builtin_floorf(x); }
MSNE_HD bool isinf_(float x) { return (f2u(x) & 0x7fffffffu) == 0x7f800000u; }

// ---------------- deterministic transcendentals ----------------
MSNE_HD void det_sincosf(float x, float& s, float& c) {
    float ax = absf(x);
    int j = (int)(ax * 1.27323954473516f);
    j = (j + 1) & ~1;
    float y = (float)j;
    float z = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float zz = z * z;
    float ps = ((-1.9515295891e-4f * zz + 8.3321608736e-3f) * zz - 1.6666654611e-1f) * zz * z + z;
    float pc = ((2.443315711809948e-5f * zz - 1.388731625493765e-3f) * zz + 4.166664568298827e-2f) * zz * zz - 0.5f * zz + 1.0f;
    int q = (j >> 1) & 3;
    float sv = (q & 1) ? pc : ps;
    float cv = (q & 1) ? ps : pc;
    if (q & 2) sv = -sv;
    if (q == 1 || q == 2) cv = -cv;
    s = x < 0.0f ? -sv : sv;
    c = cv;
}
MSNE_HD float det_tanf(float x) { float s, c; det_sincosf(x, s, c); return s / c; }
MSNE_HD float det_logf(float x) {
    uint32_t u = f2u(x);
    int e = (int)((u >> 23) & 0xff) - 126;
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
MSNE_HD float det_atanf_pos(float t) {
    float y0;
    if (t > 2.414213562373095f) { y0 = 1.5707963267948966f; t = -(1.0f / t); }
    else if (t > 0.4142135623730950f) { y0 = 0.7853981633974483f; t = (t - 1.0f) / (t + 1.0f); }
    else { y0 = 0.0f; }
    float z = t * t;
    float y = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t;
    return y0 + y;
}
MSNE_HD float det_atan2f(float y, float x) {
    float ay = absf(y), ax = absf(x);
    float a;
    if (ax == 0.0f && ay == 0.0f) return 0.0f;
    if (ax == 0.0f) a = 1.5707963267948966f;
    else a = det_atanf_pos(ay / ax);
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return y < 0.0f ? -a : a;
}
MSNE_HD float det_asinf_small(float x) {
    float z = x * x;
    return ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
            + 1.6666752422e-1f) * z * x + x;
}
MSNE_HD float det_acosf(float x) {
    if (x > 1.0f) x = 1.0f;
    if (x < -1.0f) x = -1.0f;
    if (x > 0.5f) return 2.0f * det_asinf_small(sqrt_(0.5f * (1.0f - x)));
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * det_asinf_small(sqrt_(0.5f * (1.0f + x)));
    return 1.5707963267948966f - det_asinf_small(x);
}

// ---------------- random.hlsl:8-46 ----------------
MSNE_HD uint32_t hash_lcg(uint32_t a) { return a * 747796405u + 2891336453u; }
MSNE_HD uint32_t hash_rxs_m_xs(uint32_t a) { uint32_t b = ((a >> ((a >> 28u) + 4u)) ^ a) * 277803737u; return (b >> 22u) ^ b; }
MSNE_HD uint32_t hash_pcg(uint32_t a) { return hash_rxs_m_xs(hash_lcg(a)); }
MSNE_HD uint32_t rng_seed(uint32_t sx, uint32_t sy, uint32_t sz) { return hash_pcg(sx + hash_pcg(sy + hash_pcg(sz))); }
MSNE_HD float rng_float(uint32_t& state) {
    state = hash_lcg(state);
    uint32_t h = hash_rxs_m_xs(state);
    return (float)(h >> 8) * 0x1p-24f;
}

// ---------------- math.hlsl ----------------
MSNE_HD float luminance(f3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }       // :17-21
MSNE_HD f3 face_forward(f3 n, f3 d) { return dot(n, d) > 0.0f ? n : neg(n); }                   // :23-25
MSNE_HD float offset_component(float p, float n) {                                            // :32-42
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int32_t of_i = (int32_t)(n * int_scale);
    int32_t pi = (int32_t)f2u(p);
    float p_i = u2f((uint32_t)(pi + (p < 0.0f ? -of_i : of_i)));
    return absf(p) < origin ? p + n * float_scale : p_i;
}
MSNE_HD f3 offset_along_normal(f3 p, f3 n) { return F3(offset_component(p.x, n.x), offset_component(p.y, n.y), offset_component(p.z, n.z)); }
MSNE_HD void coordinate_system(f3 v1, f3& v2o, f3& v3o) {                                      // :56-64
    if (absf(v1.x) > absf(v1.y)) v2o = divs(F3(-v1.z, 0.0f, v1.x), sqrt_(v1.x * v1.x + v1.z * v1.z));
    else v2o = divs(F3(0.0f, v1.z, -v1.y), sqrt_(v1.y * v1.y + v1.z * v1.z));
    v3o = cross(v2o, v1);
}

// ---------------- mappings.hlsl ----------------
MSNE_HD f2 square_to_triangle(f2 sq) { float a = 1.0f - sqrt_(1.0f - sq.x); float b = sq.y * sqrt_(1.0f - sq.x); return F2(a, b); } // :5-9
MSNE_HD f2 square_to_gaussian(f2 sq) {                                                        // :11-17
    const float u1 = 1.0f - sq.x, u2 = sq.y;
    const float r = sqrt_(-2.0f * det_logf(u1));
    const float theta = 2.0f * PI * u2;
    float s, c; det_sincosf(theta, s, c);
    return F2(r * c, r * s);
}
MSNE_HD f2 square_to_uniform_disk_concentric(f2 sq) {                                         // :19-37
    f2 o = F2(2.0f * sq.x - 1.0f, 2.0f * sq.y - 1.0f);
    if (o.x == 0.0f && o.y == 0.0f) return F2(0.0f, 0.0f);
    float theta, r;
    if (absf(o.x) > absf(o.y)) { r = o.x; theta = (PI / 4.0f) * (o.y / o.x); }
    else { r = o.y; theta = (PI / 2.0f) - (PI / 4.0f) * (o.x / o.y); }
    float s, c; det_sincosf(theta, s, c);
    return F2(r * c, r * s);
}
MSNE_HD f3 square_to_cosine_hemisphere(f2 sq) {                                               // :39-44
    f2 d = square_to_uniform_disk_concentric(sq);
    float z = sqrt_(maxf(0.0f, 1.0f - (d.x * d.x + d.y * d.y)));
    return F3(d.x, d.y, z);
}
MSNE_HD f3 spherical_to_cartesian(float sinTheta, float cosTheta, float phi) {                // :53-55
    float s, c; det_sincosf(phi, s, c);
    return F3(sinTheta * c, sinTheta * s, cosTheta);
}
MSNE_HD f2 cartesian_to_spherical(f3 v) {                                                     // :59-64
    float p = det_atan2f(v.y, v.x);
    float phi = (p < 0.0f) ? (p + 2.0f * PI) : p;
    float theta = det_acosf(v.z);
    return F2(phi, theta);
}
MSNE_HD f3 square_to_equal_area_sphere(f2 sq) {                                               // :67-83
    const f2 uv = F2(2.0f * sq.x - 1.0f, 2.0f * sq.y - 1.0f);
    const f2 uvp = F2(absf(uv.x), absf(uv.y));
    const float signedDistance = 1.0f - (uvp.x + uvp.y);
    const float d = absf(signedDistance);
    const float r = 1.0f - d;
    const float phi = (r == 0.0f ? 1.0f : (uvp.y - uvp.x) / r + 1.0f) * PI / 4.0f;
    float s, c; det_sincosf(phi, s, c);
    const float q = sqrt_(2.0f - r * r);
    return F3(signf(uv.x) * (c * r * q), signf(uv.y) * (s * r * q), signf(signedDistance) * (1.0f - r * r));
}
MSNE_HD f2 square_to_equal_area_sphere_inverse(f3 dir) {                                      // :85-99
    const f3 a = F3(absf(dir.x), absf(dir.y), absf(dir.z));
    const float r = sqrt_(1.0f - a.z);
    float phi = (a.x == 0.0f && a.y == 0.0f) ? 0.0f : det_atan2f(minf(a.x, a.y), maxf(a.x, a.y)) * 2.0f / PI;
    if (a.x < a.y) phi = 1.0f - phi;
    f2 uv = F2(r - phi * r, phi * r);
    if (dir.z < 0.0f) uv = F2(1.0f - uv.y, 1.0f - uv.x);
    uv.x *= signf(dir.x); uv.y *= signf(dir.y);
    return F2((uv.x + 1.0f) / 2.0f, (uv.y + 1.0f) / 2.0f);
}
MSNE_HD bool coin_flip_remap(float p, float& rand) {                                          // :103-111
    if (rand < p) { rand /= p; return true; }
    rand = (rand - p) / (1.0f - p);
    return false;
}

// ---------------- reflection_frame.hlsl ----------------
struct Frame { f3 n, s, t; };
MSNE_HD void frame_reorthogonalize(Frame& f) {                                                // :32-36
    f.s = normalize(sub(f.s, scale(f.n, dot(f.n, f.s))));
    f.t = normalize(cross(f.n, f.s));
}
MSNE_HD f3 frame_world_to_frame(const Frame& f, f3 v) { return F3(dot(f.s, v), dot(f.t, v), dot(f.n, v)); }   // :38-41
MSNE_HD f3 frame_frame_to_world(const Frame& f, f3 v) {                                       // :43-46
    return F3(f.s.x * v.x + f.t.x * v.y + f.n.x * v.z,
              f.s.y * v.x + f.t.y * v.y + f.n.y * v.z,
              f.s.z * v.x + f.t.z * v.y + f.n.z * v.z);
}
MSNE_HD float frame_cos2theta(f3 v) { return v.z * v.z; }
MSNE_HD float frame_sin2theta(f3 v) { return maxf(0.0f, 1.0f - frame_cos2theta(v)); }
MSNE_HD float frame_tan2theta(f3 v) { return frame_sin2theta(v) / frame_cos2theta(v); }
MSNE_HD bool frame_same_hemisphere(f3 a, f3 b) { return a.z * b.z > 0.0f; }

// ---------------- vector.zig Mat3x4 ----------------
MSNE_HD f3 m34_mul_point(const m34& m, f3 p) {                                                // vector.zig:277-282
    return F3(m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3] * 1.0f,
              m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3] * 1.0f,
              m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3] * 1.0f);
}
MSNE_HD f3 m34_mul_vec(const m34& m, f3 p) {                                                  // vector.zig:284-289
    return F3(m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z,
              m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z,
              m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z);
}
MSNE_HD f3 m34_mul_transposed(const m34& m, f3 v) {                                           // reflection_frame.hlsl:24-30
    return F3(m.m[0][0] * v.x + m.m[1][0] * v.y + m.m[2][0] * v.z,
              m.m[0][1] * v.x + m.m[1][1] * v.y + m.m[2][1] * v.z,
              m.m[0][2] * v.x + m.m[1][2] * v.y + m.m[2][2] * v.z);
}
MSNE_HD m34 m34_inverse_affine(const m34& s) {                                                // vector.zig:350-362, :512-520
    f3 px = F3(s.m[0][0], s.m[1][0], s.m[2][0]);
    f3 py = F3(s.m[0][1], s.m[1][1], s.m[2][1]);
    f3 pz = F3(s.m[0][2], s.m[1][2], s.m[2][2]);
    f3 v = F3(s.m[0][3], s.m[1][3], s.m[2][3]);
    float det = dot(px, cross(py, pz));
    float inv = 1.0f / det;
    f3 v1 = scale(cross(py, pz), inv), v2 = scale(cross(pz, px), inv), v3 = scale(cross(px, py), inv);
    f3 ix = F3(v1.x, v2.x, v3.x), iy = F3(v1.y, v2.y, v3.y), iz = F3(v1.z, v2.z, v3.z);
    f3 nx = scale(ix, -1.0f), ny = scale(iy, -1.0f), nz = scale(iz, -1.0f);
    f3 res = scale(nx, v.x);
    res = add(scale(ny, v.y), res);
    res = add(scale(nz, v.z), res);
    m34 o;
    o.m[0][0] = ix.x; o.m[0][1] = iy.x; o.m[0][2] = iz.x; o.m[0][3] = res.x;
    o.m[1][0] = ix.y; o.m[1][1] = iy.y; o.m[1][2] = iz.y; o.m[1][3] = res.y;
    o.m[2][0] = ix.z; o.m[2][1] = iy.z; o.m[2][2] = iz.z; o.m[2][3] = res.z;
    return o;
}

}  // namespace msne
