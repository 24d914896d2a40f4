// valu_microbench.hip — calibrates the VALU issue ceiling of one gfx950 SIMD for the instruction mix of the traversal kernels'
// node body (csrc/trace.hip step_node): register-only loops, each instruction kind alone and in the node body's proportion, at
// 1..8 resident waves per SIMD.  Reports shader cycles (s_memtime) per wave-instruction per SIMD — the number that turns
// "wave-instructions per ray" into a ceiling in rays/s — and the clock the part actually ran at (s_memtime ticks / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_microbench tools/valu_microbench.hip && /tmp/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

// a block = ONE asm statement of 8 instructions, one per chain register r0..r7: 8 independent chains, so a single wave never waits for
// its own previous result, and the compiler's hazard recogniser puts no s_nop between them
#define B8(OP, TAIL) asm volatile(OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                                  OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL \
                                  : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x), "v"(y), "v"(q[0]), "v"(q[1]) : "vcc");
#define B_FMA   B8("v_fma_f32", ", %8, %9")
#define B_MUL   B8("v_mul_f32", ", %8")
#define B_MAX3  B8("v_max3_f32", ", %8, %9")
#define B_MIN3  B8("v_min3_f32", ", %8, %9")
#define B_CND   B8("v_cndmask_b32", ", %8, vcc")
#define B_AND   B8("v_and_b32", ", %10")
#define B_ADDU  B8("v_add_u32", ", %11")
#define B_RCP   B8("v_rcp_f32", "")
#define B_LSHR  asm volatile("v_lshrrev_b32 %0, 1, %0\nv_lshrrev_b32 %1, 1, %1\nv_lshrrev_b32 %2, 1, %2\nv_lshrrev_b32 %3, 1, %3\nv_lshrrev_b32 %4, 1, %4\nv_lshrrev_b32 %5, 1, %5\nv_lshrrev_b32 %6, 1, %6\nv_lshrrev_b32 %7, 1, %7" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
// sources are the four packed-byte words q0..q3, destinations the chains (like the node body: 48 bytes of planes -> 48 floats)
#define B8Q(OP, TAIL) asm volatile(OP " %0, %8" TAIL "\n" OP " %1, %9" TAIL "\n" OP " %2, %10" TAIL "\n" OP " %3, %11" TAIL "\n" \
                                   OP " %4, %8" TAIL "\n" OP " %5, %9" TAIL "\n" OP " %6, %10" TAIL "\n" OP " %7, %11" TAIL \
                                   : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]));
#define B_CVT0  B8Q("v_cvt_f32_ubyte0", "")
#define B_CVT1  B8Q("v_cvt_f32_ubyte1", "")
#define B_CVT2  B8Q("v_cvt_f32_ubyte2", "")
#define B_CVT3  B8Q("v_cvt_f32_ubyte3", "")
#define B_BFE   B8Q("v_bfe_u32", ", 8, 8")
#define B_CMP   asm volatile("v_cmp_le_f32 vcc, %0, %8\nv_cmp_le_f32 vcc, %1, %8\nv_cmp_le_f32 vcc, %2, %8\nv_cmp_le_f32 vcc, %3, %8\nv_cmp_le_f32 vcc, %4, %8\nv_cmp_le_f32 vcc, %5, %8\nv_cmp_le_f32 vcc, %6, %8\nv_cmp_le_f32 vcc, %7, %8" \
                             : : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(x) : "vcc");
// compare + select pairs as the node body has them (hit mask: v_cmp -> v_cndmask)
#define B_CMPCND asm volatile("v_cmp_le_f32 vcc, %0, %4\nv_cndmask_b32 %0, %0, %4, vcc\nv_cmp_le_f32 vcc, %1, %4\nv_cndmask_b32 %1, %1, %4, vcc\nv_cmp_le_f32 vcc, %2, %4\nv_cndmask_b32 %2, %2, %4, vcc\nv_cmp_le_f32 vcc, %3, %4\nv_cndmask_b32 %3, %3, %4, vcc" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(x) : "vcc");
#define B_PKFMA asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\nv_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4" \
                             : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(px));
#define B_DEP   asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2" \
                             : "+v"(r[0]) : "v"(x), "v"(y));   // ONE chain: dependent-issue latency
#define B_MAX   B8("v_max_f32", ", %8")
#define B_MIN   B8("v_min_f32", ", %8")
#define B_SUB   B8("v_sub_f32", ", %8")
#define B_MED3  B8("v_med3_f32", ", %8, %9")
#define B_LSHLOR B8("v_lshl_or_b32", ", 1, %10")
#define B_ANDOR B8("v_and_or_b32", ", %10, %11")
#define B_ALIGN B8("v_alignbit_b32", ", %10, %11")
#define B_BFI   B8("v_bfi_b32", ", %10, %11")
#define B_MAD24 B8("v_mad_u32_u24", ", %10, %11")
#define B_PERM  B8Q("v_perm_b32", ", %9, %10")
#define B_CVTU  B8Q("v_cvt_f32_u32", "")
#define B_CVTH  B8Q("v_cvt_f32_f16", "")
// v_fma_mix_f32: f32 = fma(f16 half of a packed register, f32, f32) — conversion and fma in one instruction (lo half / hi half)
#define B8M(SEL) asm volatile("v_fma_mix_f32 %0, %8, %12, %13 " SEL "\nv_fma_mix_f32 %1, %9, %12, %13 " SEL "\nv_fma_mix_f32 %2, %10, %12, %13 " SEL "\nv_fma_mix_f32 %3, %11, %12, %13 " SEL "\n" \
                              "v_fma_mix_f32 %4, %8, %12, %13 " SEL "\nv_fma_mix_f32 %5, %9, %12, %13 " SEL "\nv_fma_mix_f32 %6, %10, %12, %13 " SEL "\nv_fma_mix_f32 %7, %11, %12, %13 " SEL \
                              : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(x), "v"(y));
#define B_MIXLO B8M("op_sel_hi:[1,0,0]")
#define B_MIXHI B8M("op_sel:[1,0,0] op_sel_hi:[1,0,0]")
#define B_PKFMAH B8("v_pk_fma_f16", ", %10, %11")
#define B_PKMINH B8("v_pk_min_f16", ", %10")
// compare into an SGPR pair / select by an SGPR pair (VOP3 forms, what hipcc emits for the hit mask)
#define B_CMPS  asm volatile("v_cmp_le_f32 s[20:21], %0, %8\nv_cmp_le_f32 s[22:23], %1, %8\nv_cmp_le_f32 s[20:21], %2, %8\nv_cmp_le_f32 s[22:23], %3, %8\nv_cmp_le_f32 s[20:21], %4, %8\nv_cmp_le_f32 s[22:23], %5, %8\nv_cmp_le_f32 s[20:21], %6, %8\nv_cmp_le_f32 s[22:23], %7, %8" \
                             : : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(x) : "s20", "s21", "s22", "s23");
#define B_CNDS  asm volatile("v_cndmask_b32 %0, %0, %8, %9\nv_cndmask_b32 %1, %1, %8, %9\nv_cndmask_b32 %2, %2, %8, %9\nv_cndmask_b32 %3, %3, %8, %9\nv_cndmask_b32 %4, %4, %8, %9\nv_cndmask_b32 %5, %5, %8, %9\nv_cndmask_b32 %6, %6, %8, %9\nv_cndmask_b32 %7, %7, %8, %9" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x), "s"(smask));
// cndmask with DISTINCT destination and sources (the in-place vcc form above ran at 23 cycles: is it the form or the instruction?)
#define B_CNDD  asm volatile("v_cndmask_b32 %0, %8, %9, vcc\nv_cndmask_b32 %1, %9, %8, vcc\nv_cndmask_b32 %2, %8, %9, vcc\nv_cndmask_b32 %3, %9, %8, vcc\nv_cndmask_b32 %4, %8, %9, vcc\nv_cndmask_b32 %5, %9, %8, vcc\nv_cndmask_b32 %6, %8, %9, vcc\nv_cndmask_b32 %7, %9, %8, vcc" \
                             : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(x), "v"(y));
// the candidate node body: planes as packed f16 -> v_fma_mix_f32, 24 swaps (v_alignbit) instead of 12 selects, sign-bit hit mask (v_sub + v_lshl_or) instead of cmp + cndmask
#define B_LDS   asm volatile("ds_read_b32 %0, %8\nds_read_b32 %1, %8 offset:1024\nds_read_b32 %2, %8 offset:2048\nds_read_b32 %3, %8 offset:3072\nds_read_b32 %4, %8 offset:4096\nds_read_b32 %5, %8 offset:5120\nds_read_b32 %6, %8 offset:6144\nds_read_b32 %7, %8 offset:7168\ns_waitcnt lgkmcnt(0)" \
                             : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(ldsaddr));
#define B_ANDB  B8("v_and_b32", ", %10")
#define B_ORB   B8("v_or_b32", ", %10")
#define B_XORB  B8("v_xor_b32", ", %10")
#define B_ADDF  B8("v_add_f32", ", %8")
#define B_LSHL  asm volatile("v_lshlrev_b32 %0, 1, %0\nv_lshlrev_b32 %1, 1, %1\nv_lshlrev_b32 %2, 1, %2\nv_lshlrev_b32 %3, 1, %3\nv_lshlrev_b32 %4, 1, %4\nv_lshlrev_b32 %5, 1, %5\nv_lshlrev_b32 %6, 1, %6\nv_lshlrev_b32 %7, 1, %7" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
#define B_MOV   B8Q("v_mov_b32", "")
// SDWA: byte n of a packed word -> byte 1 of the destination, the other destination bytes preserved (0x3F80xx00: the float 1 + q / 32768)
#define B_MOVSDWA asm volatile("v_mov_b32_sdwa %0, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nv_mov_b32_sdwa %1, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n" \
                               "v_mov_b32_sdwa %2, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nv_mov_b32_sdwa %3, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n" \
                               "v_mov_b32_sdwa %4, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nv_mov_b32_sdwa %5, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n" \
                               "v_mov_b32_sdwa %6, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nv_mov_b32_sdwa %7, %9 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" \
                               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(q[0]), "v"(q[1]));
#define B_ORSDWA asm volatile("v_or_b32_sdwa %0, %8, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\nv_or_b32_sdwa %1, %8, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n" \
                              "v_or_b32_sdwa %2, %8, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\nv_or_b32_sdwa %3, %8, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n" \
                              "v_or_b32_sdwa %4, %9, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\nv_or_b32_sdwa %5, %9, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n" \
                              "v_or_b32_sdwa %6, %9, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\nv_or_b32_sdwa %7, %9, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" \
                              : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(q[0]), "v"(q[1]), "v"(q[2]));
#define B_CVTSDWA asm volatile("v_cvt_f32_u32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\nv_cvt_f32_u32_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n" \
                               "v_cvt_f32_u32_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\nv_cvt_f32_u32_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n" \
                               "v_cvt_f32_u32_sdwa %4, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\nv_cvt_f32_u32_sdwa %5, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n" \
                               "v_cvt_f32_u32_sdwa %6, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\nv_cvt_f32_u32_sdwa %7, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" \
                               : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(q[0]), "v"(q[1]));
// the vcc form of v_cndmask ran at 23 cycles above with a vcc that no instruction of the loop writes: is it vcc, or who wrote it last?
#define B_CNDV  asm volatile("s_mov_b64 vcc, %9\nv_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x), "s"(smask) : "vcc");
#define B_CNDW  asm volatile("v_cmp_le_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x) : "vcc");
#define B_CNDE64 asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\nv_cndmask_b32_e64 %1, %1, %8, vcc\nv_cndmask_b32_e64 %2, %2, %8, vcc\nv_cndmask_b32_e64 %3, %3, %8, vcc\nv_cndmask_b32_e64 %4, %4, %8, vcc\nv_cndmask_b32_e64 %5, %5, %8, vcc\nv_cndmask_b32_e64 %6, %6, %8, vcc\nv_cndmask_b32_e64 %7, %7, %8, vcc" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x));
// what compiled code looks like: one compare, one select on its result, six other instructions
#define B_SELV  asm volatile("v_cmp_le_f32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x), "v"(y) : "vcc");
#define B_SELS  asm volatile("v_cmp_le_f32 s[20:21], %0, %8\nv_cndmask_b32 %1, %1, %8, s[20:21]\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9" \
                             : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(x), "v"(y) : "s20", "s21");
#define X8(B) B B B B B B B B

#define KINDS(X) \
    X(FMA, "v_fma_f32", 64, X8(B_FMA)) X(MUL, "v_mul_f32", 64, X8(B_MUL)) X(SUB, "v_sub_f32", 64, X8(B_SUB)) \
    X(CVT, "v_cvt_f32_ubyteN", 64, B_CVT0 B_CVT1 B_CVT2 B_CVT3 B_CVT0 B_CVT1 B_CVT2 B_CVT3) X(CVTU, "v_cvt_f32_u32", 64, X8(B_CVTU)) X(CVTH, "v_cvt_f32_f16", 64, X8(B_CVTH)) \
    X(MIXLO, "v_fma_mix_f32 (f16 lo half, f32, f32)", 64, X8(B_MIXLO)) X(MIXHI, "v_fma_mix_f32 (f16 hi half, f32, f32)", 64, X8(B_MIXHI)) \
    X(MINMAX3, "v_min3/max3_f32", 64, B_MAX3 B_MIN3 B_MAX3 B_MIN3 B_MAX3 B_MIN3 B_MAX3 B_MIN3) X(MINMAX, "v_min/max_f32 (two operands)", 64, B_MAX B_MIN B_MAX B_MIN B_MAX B_MIN B_MAX B_MIN) X(MED3, "v_med3_f32", 64, X8(B_MED3)) \
    X(CMP, "v_cmp_le_f32 vcc", 64, X8(B_CMP)) X(CMPS, "v_cmp_le_f32 s[n:n+1]", 64, X8(B_CMPS)) \
    X(CND, "v_cndmask_b32 vcc, in place", 64, X8(B_CND)) X(CNDD, "v_cndmask_b32 vcc, distinct registers", 64, X8(B_CNDD)) X(CNDS, "v_cndmask_b32 s[n:n+1]", 64, X8(B_CNDS)) X(CNDE64, "v_cndmask_b32_e64 ... vcc (VOP3 encoding)", 64, X8(B_CNDE64)) X(SELV, "v_cmp vcc + v_cndmask vcc + 6 v_fma", 64, X8(B_SELV)) X(SELS, "v_cmp s[20:21] + v_cndmask s[20:21] + 6 v_fma", 64, X8(B_SELS)) X(CNDV, "s_mov_b64 vcc + 8 v_cndmask_b32 vcc", 72, X8(B_CNDV)) X(CNDW, "v_cmp vcc + 8 v_cndmask_b32 vcc", 72, X8(B_CNDW)) \
    X(ANDB, "v_and_b32", 64, X8(B_ANDB)) X(ORB, "v_or_b32", 64, X8(B_ORB)) X(XORB, "v_xor_b32", 64, X8(B_XORB)) X(LSHR, "v_lshrrev_b32", 64, X8(B_LSHR)) X(LSHL, "v_lshlrev_b32", 64, X8(B_LSHL)) \
    X(ADDU, "v_add_u32", 64, X8(B_ADDU)) X(BFE, "v_bfe_u32", 64, X8(B_BFE)) X(MOV, "v_mov_b32", 64, X8(B_MOV)) X(ADDF, "v_add_f32", 64, X8(B_ADDF)) \
    X(MOVSDWA, "v_mov_b32_sdwa byte n -> byte 1, preserve", 64, X8(B_MOVSDWA)) X(ORSDWA, "v_or_b32_sdwa byte n | magic", 64, X8(B_ORSDWA)) X(CVTSDWA, "v_cvt_f32_u32_sdwa byte n", 64, X8(B_CVTSDWA)) \
    X(INTOPS, "int (and/lshr/bfe/add)", 64, B_AND B_LSHR B_BFE B_ADDU B_AND B_LSHR B_BFE B_ADDU) X(PERM, "v_perm_b32", 64, X8(B_PERM)) X(LSHLOR, "v_lshl_or_b32", 64, X8(B_LSHLOR)) X(ANDOR, "v_and_or_b32", 64, X8(B_ANDOR)) \
    X(ALIGN, "v_alignbit_b32", 64, X8(B_ALIGN)) X(BFI, "v_bfi_b32", 64, X8(B_BFI)) X(MAD24, "v_mad_u32_u24", 64, X8(B_MAD24)) \
    X(PKFMA, "v_pk_fma_f32", 64, X8(B_PKFMA)) X(PKFMAH, "v_pk_fma_f16", 64, X8(B_PKFMAH)) X(PKMINH, "v_pk_min_f16", 64, X8(B_PKMINH)) X(RCP, "v_rcp_f32", 64, X8(B_RCP)) \
    X(DEPFMA, "v_fma_f32 dependent chain", 64, X8(B_DEP)) X(LDS, "ds_read_b32 x8 + waitcnt (latency chain)", 64, X8(B_LDS)) \
    /* the node body of k_trace_closest<false,false>: 48 cvt, 48 fma, 32 min3/max3, 16 mul, 8 cmp, 8 + 16 cndmask, 32 integer = 208 */ \
    X(NODEMIX, "node body today: 48cvt 48fma 32mm3 16mul 8cmp 24cnd 32int", 208, B_CND B_CND B_MUL B_MUL B_AND B_LSHR B_BFE B_ADDU \
      B_CVT0 B_FMA B_CVT1 B_FMA B_CVT2 B_FMA B_CVT3 B_FMA B_CVT0 B_FMA B_CVT1 B_FMA B_MAX3 B_MIN3 B_MAX3 B_MIN3 B_CMPCND B_CMPCND) \
    /* candidate: byte planes -> 24 v_perm (two f16 = 1024 + q per word) -> 48 v_fma_mix, same min3/max3, 16 mul, 16 cnd, 8 sub + 8 lshl_or for the hit mask, 32 int = 184 */ \
    X(NODEMIX2, "node body candidate: 24perm 48fma_mix 32mm3 16mul 16cnd 8sub 8lshl_or 32int", 184, B_CND B_CND B_MUL B_MUL B_AND B_LSHR B_BFE B_ADDU \
      B_PERM B_PERM B_PERM B_MIXLO B_MIXHI B_MIXLO B_MIXHI B_MIXLO B_MIXHI B_MAX3 B_MIN3 B_MAX3 B_MIN3 B_SUB B_LSHLOR)
#define X_ENUM(id, name, n, body) id,
#define X_NAME(id, name, n, body) name,
#define X_INST(id, name, n, body) n,
#define X_BODY(id, name, n, body) if (KIND == id) { body }
#define X_KERN(id, name, n, body) k_bench<id>,
enum Kind { KINDS(X_ENUM) N_KIND };
static const char* kind_name[N_KIND] = { KINDS(X_NAME) };
static const int kind_insts[N_KIND] = { KINDS(X_INST) };   // wave-instructions per loop iteration

template <int KIND>
__global__ __launch_bounds__(256) void k_bench(float* out, unsigned long long* cyc, unsigned long long* real, int iters, float xin, uint32_t qin, unsigned long long lane_mask) {
    float r[8]; uint32_t q[4]; typedef float f2 __attribute__((ext_vector_type(2))); f2 p[4]; f2 px = { xin, xin };
    const float x = xin, y = xin * 0.5f;
    const unsigned long long smask = 0x5555555555555555ull * (unsigned long long)(qin & 3u);
    __shared__ uint32_t lds[2048 + 256]; lds[threadIdx.x] = qin; const uint32_t ldsaddr = (uint32_t)(threadIdx.x * 4);
    for (int k = 0; k < 8; k++) r[k] = xin + (float)(threadIdx.x + k);
    for (int k = 0; k < 4; k++) { q[k] = qin + threadIdx.x * 0x01010101u + k; p[k] = f2{ r[k], r[k + 4] }; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if ((lane_mask >> (threadIdx.x & 63)) & 1ull)   // partial waves: does an instruction cost less when few lanes are enabled?
    for (int it = 0; it < iters; it++) {
        KINDS(X_BODY)
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0; for (int k = 0; k < 8; k++) s += r[k]; for (int k = 0; k < 4; k++) s += p[k].x + p[k].y;
    if (s == 12345.678f) out[0] = s;   // keep the chains alive
    if ((threadIdx.x & 63) == 0) { atomicMax(cyc, t1 - t0); atomicMax(real, w1 - w0); }
}

typedef void (*KFn)(float*, unsigned long long*, unsigned long long*, int, float, uint32_t, unsigned long long);
static KFn kernels[N_KIND] = { KINDS(X_KERN) };

int main() {
    hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    float* out; unsigned long long* d; hipMalloc(&out, 4); hipMalloc(&d, 16);
    printf("# %s, %d CUs, clockRate %d kHz.  Blocks of 256 threads = one wave per SIMD each; W blocks per CU = W resident waves per SIMD.\n", prop.gcnArchName, cus, prop.clockRate);
    printf("# cyc/inst = slowest wave's s_memtime ticks / (W x wave-instructions per wave): SIMD cycles per issued wave-instruction.\n");
    printf("%-80s %3s %10s %9s %9s %8s\n", "instruction", "W", "cyc/inst", "GHz", "ms", "Ginst/s");
    const int iters = 20000;
    const int ws[] = { 1, 2, 4, 6, 8 };   // (the summary lines of tools/run_valu_microbench.sh show W = 2 and 6)
    for (int k = 0; k < N_KIND; k++) for (int wi = 0; wi < 5; wi++) {
        const int W = ws[wi];
        double best = 1e30, ghz = 0, msb = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(d, 0, 16);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kernels[k], dim3(cus * W), dim3(256), 0, 0, out, d, d + 1, iters, 1.0001f, 0x01020304u, ~0ull);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            const double per = (double)h[0] / ((double)W * iters * kind_insts[k]);
            if (per < best) { best = per; ghz = (double)h[0] / ((double)h[1] * 10.0); msb = ms; }   // s_memrealtime: 100 MHz
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        const double ginst = (double)cus * 4 * W * iters * kind_insts[k] / (msb * 1e-3) / 1e9;
        printf("%-80s %3d %10.3f %9.3f %9.3f %8.1f\n", kind_name[k], W, best, ghz, msb, ginst);
    }
    printf("# partial waves (W = 6): the same loops with only some lanes enabled\n");
    const struct { const char* name; unsigned long long mask; } masks[] = { { "64 lanes", ~0ull }, { "lanes 0-31", 0xffffffffull }, { "lanes 0-15", 0xffffull }, { "every 4th lane (16)", 0x1111111111111111ull }, { "lane 0 only", 1ull } };
    const int pk[] = { FMA, CVT, MINMAX3, NODEMIX };
    for (int k : pk) for (auto& m : masks) {
        const int W = 6; float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(d, 0, 16);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kernels[k], dim3(cus * W), dim3(256), 0, 0, out, d, d + 1, iters, 1.0001f, 0x01020304u, m.mask);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        printf("%-80s %-22s %9.3f ms %8.1f Ginst/s\n", kind_name[k], m.name, best, (double)cus * 4 * W * iters * kind_insts[k] / (best * 1e-3) / 1e9);
    }
    return 0;
}
