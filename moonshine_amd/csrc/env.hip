// env.hip — environment-map preprocessing, the three compute kernels of shaders/background/ run by
// BackgroundManager.addBackground (engine/hrtsystem/BackgroundManager.zig:142-394):
//   k_env_equirect_to_equal_area  ↔ equirectangular_to_equal_area.hlsl:9-30  (9-tap resample, mirrored-repeat bilinear source)
//   k_env_luminance               ↔ luminance.hlsl:7-15
//   k_env_fold                    ↔ fold.hlsl:6-17  (2x2 SUM, so the 1x1 top level is the integral)
// Same 8x8 workgroup shape as the reference; expression order matches the test oracle.
#include "shade.h"

namespace msne {

__global__ __launch_bounds__(64) void k_env_equirect_to_equal_area(const float4* src, uint32_t sw, uint32_t sh_, float4* dst, uint32_t S) {
    const uint32_t px = blockIdx.x * 8 + threadIdx.x, py = blockIdx.y * 8 + threadIdx.y;
    if (px >= S || py >= S) return;
    f3 color = F3(0.0f, 0.0f, 0.0f);
    const uint32_t spd = 3;
    for (uint32_t i = 0; i < spd; i++) for (uint32_t j = 0; j < spd; j++) {
        const f2 sub_ = F2((float)(1 + i) / (float)(spd + 1), (float)(1 + j) / (float)(spd + 1));
        const f2 dc = F2(((float)px + sub_.x) / (float)S, ((float)py + sub_.y) / (float)S);
        const f3 dir = square_to_equal_area_sphere(dc);
        const f2 sph = cartesian_to_spherical(dir);
        const f2 sc = F2(sph.x / (2.0f * PI), sph.y / PI);
        const float4 o = sample_bilinear(src, sw, sh_, sc.x, sc.y, true);
        color = add(color, F3(o.x, o.y, o.z));
    }
    dst[(size_t)py * S + px] = make_float4(color.x / 9.0f, color.y / 9.0f, color.z / 9.0f, 1.0f);
}

__global__ __launch_bounds__(64) void k_env_luminance(const float4* rgb, float* lum0, uint32_t S) {
    const uint32_t px = blockIdx.x * 8 + threadIdx.x, py = blockIdx.y * 8 + threadIdx.y;
    if (px >= S || py >= S) return;
    const float4 c = rgb[(size_t)py * S + px];
    lum0[(size_t)py * S + px] = luminance(F3(c.x, c.y, c.z));
}

__global__ __launch_bounds__(64) void k_env_fold(const float* srcm, float* dstm, uint32_t d) {
    const uint32_t x = blockIdx.x * 8 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
    if (x >= d || y >= d) return;
    const uint32_t sd = d * 2;
    dstm[(size_t)y * d + x] = srcm[(size_t)(2 * y) * sd + 2 * x] + srcm[(size_t)(2 * y) * sd + 2 * x + 1]
                            + srcm[(size_t)(2 * y + 1) * sd + 2 * x] + srcm[(size_t)(2 * y + 1) * sd + 2 * x + 1];
}

// one float4 per 2x2 quad of a level: the four texels one step of the mip descent (light.hlsl:52-66) reads
__global__ __launch_bounds__(64) void k_env_quads(const float* level, float4* quads, uint32_t d /* quads per side = level size / 2 */) {
    const uint32_t x = blockIdx.x * 8 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
    if (x >= d || y >= d) return;
    const uint32_t sd = d * 2;
    quads[(size_t)y * d + x] = make_float4(level[(size_t)(2 * y) * sd + 2 * x], level[(size_t)(2 * y + 1) * sd + 2 * x],
                                           level[(size_t)(2 * y) * sd + 2 * x + 1], level[(size_t)(2 * y + 1) * sd + 2 * x + 1]);
}
void launch_env_quads(hipStream_t s, const float* lum, const uint32_t* lum_offset, float4* quads, const uint32_t* quad_offset, uint32_t S, uint32_t mips) {
    for (uint32_t l = 0; l + 1 < mips; l++) {
        const uint32_t d = (S >> l) / 2;
        hipLaunchKernelGGL(k_env_quads, dim3((d + 7) / 8, (d + 7) / 8, 1), dim3(8, 8, 1), 0, s, lum + lum_offset[l], quads + quad_offset[l], d);
    }
}

void launch_env_build(hipStream_t s, const float4* src, uint32_t sw, uint32_t sh_, float4* rgb, float* lum, const uint32_t* lum_offset, uint32_t S, uint32_t mips) {
    const dim3 blk(8, 8, 1);
    const dim3 grid((S + 7) / 8, (S + 7) / 8, 1);
    hipLaunchKernelGGL(k_env_equirect_to_equal_area, grid, blk, 0, s, src, sw, sh_, rgb, S);
    hipLaunchKernelGGL(k_env_luminance, grid, blk, 0, s, rgb, lum + lum_offset[0], S);
    for (uint32_t l = 1; l < mips; l++) {
        const uint32_t d = S >> l;
        hipLaunchKernelGGL(k_env_fold, dim3((d + 7) / 8, (d + 7) / 8, 1), blk, 0, s, lum + lum_offset[l - 1], lum + lum_offset[l], d);
    }
}

}  // namespace msne
