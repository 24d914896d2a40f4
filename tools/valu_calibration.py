"""profiles/r03_valu_microbench.txt (tools/valu_microbench.hip on one MI355X) -> profiles/r03_valu_calibration.json: measured SIMD cycles per
wave-instruction at 6 waves per SIMD (= 1024 SIMDs x measured GHz / measured Ginst/s) for every instruction tried, and the per-class issue
costs bench.py's VALU roofline uses.   python tools/valu_calibration.py [gpurun_out/r03_valu_microbench.txt]"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03_valu_microbench.txt")
dst = os.path.join(ROOT, "profiles", "r03_valu_microbench.txt")
if os.path.abspath(src) != dst:
    shutil.copyfile(src, dst)
meas = {}
for line in open(dst):
    m = re.match(r"^(.{80})\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip("\n"))
    if m and int(m.group(2)) == 6:
        ghz, ginst = float(m.group(4)), float(m.group(6))
        meas[m.group(1).strip()] = {"ginst_per_s": ginst, "ghz": ghz, "cycles_per_wave_instruction": round(1024.0 * ghz / ginst, 3)}
out = {
    "source": "profiles/r03_valu_microbench.txt (tools/valu_microbench.hip: register-only loops of 64 independent instructions, 6 waves per SIMD, whole chip)",
    "measured_at_6_waves_per_simd": meas,
    # what the table says: a gfx950 SIMD issues f32 fma / mul / add / sub and the simple integer ops (and, or, xor, lshrrev, add_u32, mov) in 2 cycles per
    # wave64 instruction and EVERYTHING else tried in 4 (conversions incl. v_cvt_f32_ubyteN, min / max / med3, compares, v_cndmask, v_perm, v_lshlrev, v_bfe,
    # every three-operand integer op, v_fma_mix, packed f32 / f16 math, SDWA forms), v_rcp_f32 in 8.  The measured values sit 3-15 % above 2 / 4 / 8
    # (loop overhead: ~5 s_nop + 3 scalar instructions per 64; v_fma throttles the clock to ~2.0 GHz).  bench.py prices a kernel's instruction
    # classes (SQ_INSTS_VALU_* per ray from the committed PMC pass) at the NOMINAL costs below and INT32 at its cheaper value: a lower bound of the
    # issue cycles the kernel needs, against 1024 SIMDs x 2.4 GHz.
    "class_cycles": {"FMA_F32": 2, "MUL_F32": 2, "ADD_F32": 2, "INT32": 2, "CVT": 4, "TRANS_F32": 8, "OTHER": 4},
    "peak_simd_cycles_per_s": 1024 * 2.4e9,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_valu_calibration.json"), "w"), indent=1)
for k, v in meas.items():
    print("%-80s %6.2f cycles" % (k, v["cycles_per_wave_instruction"]))
