// gather_microbench.hip — what the memory pipeline of one gfx950 CU delivers for the traversal kernels' access pattern: every lane of a
// wave fetches its OWN record (a BVH node or a triangle), records scattered over a pool of a given size.  Round 6: the question the round-5
// verdict asked — is the L1's tag rate what binds k_trace_closest (66 tag accesses per ray = 0.86 per clock per CU), and does a fetch that needs
// fewer lookups per node visit beat today's five 16-B loads per lane?
//
//   FETCH   lane : every lane loads its record with NLOADS x global_load_dwordx4 (today's step_node: 5 x 16 B at an 80-B stride)
//           coop : the wave fetches its 64 records TOGETHER — piece q = r * 64 + lane of the wave's 64 * NLOADS pieces belongs to record q / NLOADS
//                  (ds_bpermute hands lane the index of that record), so the lanes of a quad read 64 CONTIGUOUS bytes (one tag lookup instead of four);
//                  the pieces go through an LDS staging area (ds_write_b128) and every lane reads its record back (NLOADS x ds_read_b128)
//           dlds : the same with global_load_lds_dwordx4 (gfx950: the load writes LDS itself, M0 + lane * 16 — no VGPR, no ds_write)
//   SPREAD  rand : every lane its own record anywhere in the pool (an incoherent bounce)
//           g8   : the lanes of a wave pick among 8 records of a run of 16 consecutive ones (camera / shadow rays walking the same subtree)
//           uni  : all lanes of all waves the same record (rounds 3-5's "dep" rows: the pool was one constant, so every index was the same)
//   The next index always depends on the loaded data (a traversal step: one hop at a time per lane).
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/gather_microbench tools/gather_microbench.hip && /tmp/gather_microbench
//   counters: rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace ... -- /tmp/gather_microbench pmc
//   (`pmc`: one launch per variant at W = 6, pools of 16 KiB (L1-resident) and 2 MiB (L2-resident): tools/gather_counters.py makes the table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

enum { F_LANE = 0, F_COOP = 1, F_DLDS = 2 };
enum { S_RAND = 0, S_G8 = 1, S_UNI = 2 };

template <int NLOADS, int STRIDE, int FETCH, int SPREAD>
__global__ __launch_bounds__(256) void k_gather(const uint8_t* __restrict__ pool, uint32_t nrec, int iters, uint32_t* out, unsigned long long* cyc) {
    __shared__ uint4 stage[FETCH == F_LANE ? 1 : 4 * 64 * NLOADS];
    const uint32_t gtid = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t idx = hash32(gtid) % nrec, acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        uint4 w[NLOADS];
        if (FETCH == F_LANE) {
            const uint4* p = reinterpret_cast<const uint4*>(pool + (size_t)idx * STRIDE);
#pragma unroll
            for (int k = 0; k < NLOADS; k++) w[k] = p[k];
        } else {
            uint4* st = stage + wave * 64 * NLOADS;
#pragma unroll
            for (int r = 0; r < NLOADS; r++) {
                const uint32_t q = r * 64 + lane, rec = q / NLOADS, piece = q % NLOADS;
                const uint32_t ridx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(rec * 4), (int)idx);
                const uint4* src = reinterpret_cast<const uint4*>(pool + (size_t)ridx * STRIDE) + piece;
                if (FETCH == F_DLDS) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(st + r * 64), 16, 0, 0);
                else st[q] = *src;
            }
            __builtin_amdgcn_s_waitcnt(0);            // vmcnt(0) lgkmcnt(0): the wave's own pieces are in LDS (no other wave touches this area)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < NLOADS; k++) w[k] = st[lane * NLOADS + k];
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < NLOADS; k++) x ^= w[k].x ^ w[k].y ^ w[k].z ^ w[k].w;
        acc ^= x;
        if (SPREAD == S_RAND) idx = hash32(x + it + gtid) % nrec;
        else if (SPREAD == S_G8) {      // the wave's run of 16 records moves with the data of lane 0; a lane takes one of 8 of them
            const uint32_t base = hash32((uint32_t)__builtin_amdgcn_readfirstlane((int)x) + it + (gtid >> 6)) % nrec;
            idx = (base + 2u * (hash32(lane * 31u + it) & 7u)) % nrec;
        } else idx = hash32((x & 0u) + it) % nrec;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) out[0] = acc;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}

static int g_cus = 0; static bool g_pmc = false;
static uint8_t* g_pool; static uint32_t* g_out; static unsigned long long* g_cyc; static double g_clock_ghz = 0.0;

template <int NLOADS, int STRIDE, int FETCH, int SPREAD>
static void run(const char* name) {
    static const char* fn[] = { "lane", "coop", "dlds" }; static const char* sn[] = { "rand", "g8", "uni" };
    const size_t sizes[] = { 16u << 10, 2u << 20, 16u << 20, 128u << 20 };   // a CU's L1 holds 32 KiB, one XCD's L2 4 MiB, Infinity Cache 256 MiB
    for (size_t sz : sizes) for (int W : { 4, 6 }) {
        if (g_pmc && (W != 6 || sz > (2u << 20))) continue;
        if (SPREAD == S_UNI && sz != (2u << 20)) continue;
        const uint32_t nrec = (uint32_t)(sz / STRIDE);
        const int iters = 400;
        float best = 1e30f; unsigned long long bc = 0;
        for (int rep = 0; rep < (g_pmc ? 1 : 3); rep++) {
            (void)hipMemset(g_cyc, 0, 8);
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k_gather<NLOADS, STRIDE, FETCH, SPREAD>), dim3(g_cus * W), dim3(256), 0, 0, g_pool, nrec, iters, g_out, g_cyc);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c; (void)hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
            if (ms < best) { best = ms; bc = c; }
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        const double wave_fetches = (double)g_cus * W * 4 * iters;
        printf("%-22s %-4s %-4s %6zu KiB  W=%d  %8.3f ms  %7.1f wave-fetches/us/CU  %7.1f cyc/hop/wave  %6.1f lane-loads/clk/CU(2.4)\n", name, fn[FETCH], sn[SPREAD], sz >> 10, W, best,
               wave_fetches / g_cus / (best * 1e3), (double)bc / iters, wave_fetches * 64 * NLOADS / g_cus / (best * 1e-3) / 2.4e9);
        fflush(stdout);
    }
}

template <int NLOADS, int STRIDE>
static void shape(const char* name) {
    run<NLOADS, STRIDE, F_LANE, S_RAND>(name); run<NLOADS, STRIDE, F_COOP, S_RAND>(name); run<NLOADS, STRIDE, F_DLDS, S_RAND>(name);
    run<NLOADS, STRIDE, F_LANE, S_G8>(name);   run<NLOADS, STRIDE, F_COOP, S_G8>(name);   run<NLOADS, STRIDE, F_DLDS, S_G8>(name);
    run<NLOADS, STRIDE, F_LANE, S_UNI>(name);
}

int main(int argc, char** argv) {
    g_pmc = argc > 1 && !strcmp(argv[1], "pmc");
    hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    g_cus = prop.multiProcessorCount;
    const size_t pool_bytes = (size_t)128 << 20;
    if (hipMalloc(&g_pool, pool_bytes + 256) != hipSuccess) return 1;
    (void)hipMalloc(&g_out, 4); (void)hipMalloc(&g_cyc, 8);
    {   // random content: the next index depends on it, so it must differ from record to record
        std::vector<uint32_t> h((pool_bytes + 256) / 4);
        uint32_t s = 12345u; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
        (void)hipMemcpy(g_pool, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    printf("# %s, %d CUs.  A wave-fetch = one record per lane (64 records).  W = resident waves per SIMD.  lane-loads/clk/CU at a nominal 2.4 GHz.\n", prop.gcnArchName, g_cus);
    shape<5, 80>("node 5x16B @80B");
    shape<4, 64>("node 4x16B @64B");
    shape<8, 128>("node 8x16B @128B");
    shape<3, 48>("tri 3x16B @48B");
    shape<4, 64>("tri 4x16B @64B");
    shape<1, 16>("1x16B");
    return 0;
}
