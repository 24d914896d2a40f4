// gather_microbench.hip — what the memory pipeline of one gfx950 CU delivers for the traversal kernels' access pattern: every lane of a
// wave fetches its OWN record (a BVH node or a triangle) with several 16-B loads, records scattered over a pool of a given size.
// Measures wave-level record fetches per µs per CU and the latency of one dependent hop, for the record shapes under discussion:
//   5 x 16 B at an 80-B stride (Node8 today: 2 sectors of 64 B, straddles a 128-B line half the time), 8 x 16 B at 128 B aligned,
//   4 x 16 B at 64 B aligned, 3 x 16 B at 48 B (TriRec), 2 x 16 B / 1 x 16 B.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/gather_microbench tools/gather_microbench.hip && /tmp/gather_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// DEP: the next record index depends on the loaded data (pointer chasing, one hop at a time per lane, like a traversal step);
// otherwise indices come from a hash and the loads of consecutive iterations may overlap.
template <int NLOADS, int STRIDE, bool DEP>
__global__ __launch_bounds__(256) void k_gather(const uint8_t* __restrict__ pool, uint32_t nrec, int iters, uint32_t* out, unsigned long long* cyc) {
    const uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = hash32(gtid) % nrec, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        const uint4* p = reinterpret_cast<const uint4*>(pool + (size_t)idx * STRIDE);
        uint4 w[NLOADS];
#pragma unroll
        for (int k = 0; k < NLOADS; k++) w[k] = p[k];
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < NLOADS; k++) x ^= w[k].x ^ w[k].y ^ w[k].z ^ w[k].w;
        acc ^= x;
        idx = DEP ? hash32(x + it) % nrec : hash32(gtid * 7919u + it) % nrec;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) out[0] = acc;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}

template <int NLOADS, int STRIDE, bool DEP>
static void run(const char* name, const uint8_t* pool, size_t pool_bytes, int cus, uint32_t* out, unsigned long long* cyc) {
    const size_t sizes[] = { 2u << 20, 16u << 20, 128u << 20, (size_t)2048 << 20 };   // one XCD's L2 holds 4 MiB; Infinity Cache 256 MiB
    for (size_t sz : sizes) for (int W : { 4, 6, 8 }) {
        if (sz > pool_bytes) continue;
        const uint32_t nrec = (uint32_t)(sz / STRIDE);
        const int iters = 400;
        float best = 1e30f; unsigned long long bc = 0;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipMemset(cyc, 0, 8);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k_gather<NLOADS, STRIDE, DEP>), dim3(cus * W), dim3(256), 0, 0, pool, nrec, iters, out, cyc);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            if (ms < best) { best = ms; bc = c; }
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        const double wave_fetches = (double)cus * W * 4 * iters;
        printf("%-34s %-4s %6zu MiB  W=%d  %8.3f ms  %7.1f wave-fetches/us/CU  %7.1f cyc/hop/wave  %7.1f GB/s useful  %6.1f Glane-fetches/s\n", name, DEP ? "dep" : "ind", sz >> 20, W, best,
               wave_fetches / cus / (best * 1e3), (double)bc / iters, wave_fetches * 64 * NLOADS * 16 / (best * 1e-3) / 1e9, wave_fetches * 64 / (best * 1e-3) / 1e9);
    }
}

int main() {
    hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    const size_t pool_bytes = (size_t)2048 << 20;
    uint8_t* pool; uint32_t* out; unsigned long long* cyc;
    if (hipMalloc(&pool, pool_bytes + 256) != hipSuccess) return 1;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8);
    (void)hipMemset(pool, 0x5a, pool_bytes + 256);
    printf("# %s, %d CUs.  A wave-fetch = one record per lane (64 records).  W = resident waves per SIMD.\n", prop.gcnArchName, cus);
    run<5, 80, true>("node 5x16B @80B stride (today)", pool, pool_bytes, cus, out, cyc);
    run<8, 128, true>("node 8x16B @128B aligned", pool, pool_bytes, cus, out, cyc);
    run<4, 64, true>("node 4x16B @64B aligned", pool, pool_bytes, cus, out, cyc);
    run<3, 48, true>("tri 3x16B @48B stride", pool, pool_bytes, cus, out, cyc);
    run<1, 16, true>("1x16B", pool, pool_bytes, cus, out, cyc);
    run<5, 80, false>("node 5x16B @80B stride (today)", pool, pool_bytes, cus, out, cyc);
    run<8, 128, false>("node 8x16B @128B aligned", pool, pool_bytes, cus, out, cyc);
    run<4, 64, false>("node 4x16B @64B aligned", pool, pool_bytes, cus, out, cyc);
    run<3, 48, false>("tri 3x16B @48B stride", pool, pool_bytes, cus, out, cyc);
    run<1, 16, false>("1x16B", pool, pool_bytes, cus, out, cyc);
    return 0;
}
