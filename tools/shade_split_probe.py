"""Register needs of k_shade cut in two (the round-2 verdict's item 3): derives, from csrc/integrator.hip as it is, a HIT half (classification + sort,
phase A: attributes, textures, emission, termination; phase B: queue reservation; then a 144-B hand-over record per surviving path) and a SAMPLE half
(reads the record; phase C: light samples + next direction), compiles both for gfx950 and prints the VGPR / spill counts at an unconstrained register
budget and at 128 registers (4 waves per SIMD).  Compile-time only (hipcc cross-compiles without a GPU):   python tools/shade_split_probe.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (round 3's kernel: round 4 turned k_shade into a template over path categories — profiles/r04_shade_specialised.txt — and this probe edits the source by pattern)
s = subprocess.check_output(["git", "-C", ROOT, "show", "ec77618:moonshine_amd/csrc/integrator.hip"], text=True)


def rep(old, new):
    global s
    assert old in s, old[:80]
    s = s.replace(old, new, 1)


rep('__global__ __launch_bounds__(SHADE_BLOCK, SHADE_WPS) void k_shade(SceneView sc,', 'template <int PHASE, int WPS>\n__global__ __launch_bounds__(SHADE_BLOCK, WPS) void k_shade_t(float4* handover, SceneView sc,')
rep('''        const uint32_t q = (uint32_t)(base >> 32) + (uint32_t)__popcll(mn & lt);   // sample k of this path: entry q + k * stride
''', '''        uint32_t q = (uint32_t)(base >> 32) + (uint32_t)__popcll(mn & lt);
        if (PHASE == 1) {   // the hit half ends here: what the sample half needs, 9 x 16 B per surviving path
            if (alive) {
                float4* h = handover + 9 * (size_t)j;
                h[0] = make_float4(attrs.position.x, attrs.position.y, attrs.position.z, u2f(rng));
                h[1] = make_float4(attrs.triangleFrame.n.x, attrs.triangleFrame.n.y, attrs.triangleFrame.n.z, u2f(slot));
                h[2] = make_float4(shadingFrame.n.x, shadingFrame.n.y, shadingFrame.n.z, u2f(q));
                h[3] = make_float4(shadingFrame.s.x, shadingFrame.s.y, shadingFrame.s.z, u2f(stride));
                h[4] = make_float4(shadingFrame.t.x, shadingFrame.t.y, shadingFrame.t.z, u2f(bounceCount | (delta ? 0x10000u : 0u) | (nee ? 0x20000u : 0u)));
                h[5] = make_float4(material.color.x, material.color.y, material.color.z, u2f(material.type));
                h[6] = make_float4(material.metalness, material.alpha, material.ior, 0.0f);
                h[7] = make_float4(woSs.x, woSs.y, woSs.z, 0.0f);
                h[8] = make_float4(throughput.x, throughput.y, throughput.z, 0.0f);
                lbuf[slot] = make_float4(L.x, L.y, L.z, 0.0f);
            }
            continue;
        }
''')
a = s.index('        // ---- phase C: light samples (integrator.hlsl:137-151) and the next direction (:153-165) ----\n        if (alive) {')
tail = s[a:s.index('// statistics of a finished batch')]
body = tail[tail.index('        if (alive) {'):tail.rindex('        }\n    }\n}') + len('        }\n')]
rep('''    for (uint32_t base_i = blockIdx.x * SHADE_BLOCK; base_i < n_pad; base_i += gridDim.x * SHADE_BLOCK) {
        uint32_t cat = CAT_NONE;''', '''    if (PHASE == 2) {   // the sample half: one surviving path per thread, in the order the hit half wrote them
        for (uint32_t i = blockIdx.x * SHADE_BLOCK + threadIdx.x; i < n; i += gridDim.x * SHADE_BLOCK) {
            const float4* h = handover + 9 * (size_t)i;
            const float4 h0 = h[0], h1 = h[1], h2 = h[2], h3 = h[3], h4 = h[4], h5 = h[5], h6 = h[6], h7 = h[7], h8 = h[8];
            Attrs attrs; Frame shadingFrame; Mat material;
            attrs.position = F3(h0.x, h0.y, h0.z); uint32_t rng = f2u(h0.w);
            attrs.triangleFrame.n = F3(h1.x, h1.y, h1.z); const uint32_t slot = f2u(h1.w);
            shadingFrame.n = F3(h2.x, h2.y, h2.z); const uint32_t q = f2u(h2.w);
            shadingFrame.s = F3(h3.x, h3.y, h3.z); const uint32_t stride = f2u(h3.w);
            shadingFrame.t = F3(h4.x, h4.y, h4.z); const uint32_t fl = f2u(h4.w);
            material.color = F3(h5.x, h5.y, h5.z); material.type = f2u(h5.w); material.metalness = h6.x; material.alpha = h6.y; material.ior = h6.z;
            const f3 woSs = F3(h7.x, h7.y, h7.z); f3 throughput = F3(h8.x, h8.y, h8.z);
            const float4 l4 = lbuf[slot]; f3 L = F3(l4.x, l4.y, l4.z);
            const uint32_t bounceCount = fl & 0xffffu; const bool delta = (fl & 0x10000u) != 0, nee = (fl & 0x20000u) != 0; const uint32_t j = i;
            const bool alive = true;
''' + body.replace('\n', '\n    ') + '''
        }
        return;
    }
    for (uint32_t base_i = blockIdx.x * SHADE_BLOCK; base_i < n_pad; base_i += gridDim.x * SHADE_BLOCK) {
        uint32_t cat = CAT_NONE;''')
sig = "(float4*, SceneView, PipelineOpts, PathState, HitBuf, PathState, ShadowQueue, const float4*, float4*, BounceCounters*, uint32_t);\n"
rep('void launch_shade(hipStream_t s, int grid,', "".join("template __global__ void k_shade_t<%d, %d>%s" % (ph, w, sig) for ph, w in ((0, 1), (1, 1), (2, 1), (0, 4), (1, 4), (2, 4))) + 'void launch_shade(hipStream_t s, int grid,')
rep('hipLaunchKernelGGL(k_shade, dim3(grid), dim3(SHADE_BLOCK), 0, s, sc, o,', 'hipLaunchKernelGGL((k_shade_t<0, 1>), dim3(grid), dim3(SHADE_BLOCK), 0, s, (float4*)nullptr, sc, o,')
d = tempfile.mkdtemp()
open(os.path.join(d, "integ_split.hip"), "w").write(s)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-w",
                       "-I" + os.path.join(ROOT, "moonshine_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-save-temps=obj", "-c", "integ_split.hip", "-o", "integ_split.o"], cwd=d)
asm = open(os.path.join(d, "integ_split-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
names = {0: "k_shade as shipped (one kernel)", 1: "hit half    (sort, phase A + B, writes the 144-B hand-over)", 2: "sample half (reads the hand-over, phase C)"}
print("%-62s %-28s %6s %6s" % ("kernel", "register budget", "VGPRs", "spills"))
for m in re.finditer(r"\.name:\s+_ZN4msne9k_shade_tILi(\d)ELi(\d)E.*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", asm, re.S):
    ph, w, v, sp = int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))
    print("%-62s %-28s %6d %6d" % (names[ph], "unconstrained" if w == 1 else "128 (4 waves per SIMD)", v, sp))
