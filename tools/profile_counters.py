"""Turns the scratch output of tools/profile_round.sh TAG SCENE (gpurun_out/TAG_SCENE_{trace,fetch,write,sq}) into the files
committed under profiles/ and read by bench.py:   python tools/profile_counters.py TAG SCENE

  profiles/TAG_kernel_stats_SCENE.txt   rocprofv3 --kernel-trace --stats table of the bench command + its JSON line
  profiles/TAG_counters_SCENE.json      per kernel and PER UNIT (ray / shaded path), so the figures hold at any --steps:
      hbm_bytes_per_unit                (2 x FETCH_SIZE + WRITE_SIZE) / units — FETCH_SIZE x2 is the gfx950 correction of
                                        /opt/skills/guides/MI355X_MICROARCH.md (both counters are in KB); an upper bound for 16-B gathers
      valu_wave_instructions_per_unit   SQ_INSTS_VALU / units          valu_thread_instructions_per_unit   SQ_THREAD_CYCLES_VALU / units
      lanes_per_valu_instruction        SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (of 64)
      valu_issue_busy_frac              SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x kernel time x 2.4 GHz) — NOT a utilisation: the counter books one quad-cycle
                                        per instruction (two for v_rcp) whatever it costs (tools/valu_microbench.hip under the same counters: ACTIVE / INSTS = 1.000), so it
                                        over-counts the 2-cycle f32 fma / mul / add; kept for comparison with round 2
      valu_class_per_unit               SQ_INSTS_VALU_{FMA,MUL,ADD}_F32 / INT32 / CVT / TRANS_F32 per unit (pass "valu")
      memory_pipeline                   TA / TD busy, L1 tag accesses, L1 -> L2 requests, L2 hits / misses per unit (passes "mem", "mem2")
  Counters are summed over ALL dispatches of a kernel in the run (warm-up and every repeat) and divided by the units the same run
  processed (bench.py's `profile_totals`).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, scene = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, scene))
KERNELS = ("k_trace_closest", "k_trace_shadow", "k_shade")


def bench_line(log):
    for line in open(log):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    raise SystemExit("no bench line in " + log)


def which(name):
    for k in KERNELS:
        if "msne::" + k in name:
            return k
    return None


def counters(sub):
    """-> {kernel: {counter: sum}}, {kernel: total ns}, {kernel: dispatches}"""
    val = defaultdict(lambda: defaultdict(float)); dur = defaultdict(float); seen = defaultdict(set)
    for f in glob.glob(os.path.join(G + "_" + sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = which(r["Kernel_Name"])
            if not k:
                continue
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen[k]:
                seen[k].add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return val, dur, {k: len(v) for k, v in seen.items()}


def short(n):
    return n.split("(")[0].replace("void ", "")


line = bench_line(G + "_trace.log")
cmd = open(G + "_cmd.txt").read().strip()
env_, cmd = (cmd.split(" ", 1) if cmd.startswith("MSNE_") else ("", cmd))
lines = ["# " + (env_ + " " if env_ else "") + "rocprofv3 --kernel-trace --stats --output-format csv -- python3 " + cmd + "   (MI355X)",
         "%-44s %6s %14s %12s %10s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
stats = glob.glob(os.path.join(G + "_trace", "**", "*kernel_stats.csv"), recursive=True)[0]
for r in csv.DictReader(open(stats)):
    lines.append("%-44s %6d %14d %12.0f %10d %12d %6.2f%%" % (short(r["Name"])[:44], int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]),
                                                          int(r["MinNs"]), int(r["MaxNs"]), float(r["Percentage"])))
# the table above averages over the warm-up batch (4 launches in flight) and the full batches alike; per BATCH — one launch of the kernel per bounce pass — the
# trace gives the average the bench line's roofline.avg_launch_ms is measured as
trace = glob.glob(os.path.join(G + "_trace", "**", "*kernel_trace.csv"), recursive=True)
if trace and line.get("roofline", {}).get("launches"):
    per_batch = int(line["roofline"]["launches"])
    for kern in ("k_trace_closest", "k_shade", "k_trace_shadow"):
        d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(trace[0])) if ("msne::" + kern) in r["Kernel_Name"])
        n = per_batch if kern != "k_trace_shadow" else per_batch - 1
        groups = [[v for _, v in d[g:g + n]] for g in range(0, len(d), n)]
        lines.append("# %s, batches of %d launches in trace order: %s ms in all, %s ms per launch%s" % (
            kern, n, " / ".join("%.2f" % (sum(g) / 1e6) for g in groups), " / ".join("%.3f" % (sum(g) / 1e6 / len(g)) for g in groups),
            "   <- the first is the warm-up batch; roofline.avg_launch_ms of the line below: %.3f" % line["roofline"]["avg_launch_ms"] if kern == line["roofline"].get("kernel") else ""))
lines += ["", "# the command's own line (HIP-event kernel times inside the timed region; under the profiler)", json.dumps(line)]
open(os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.txt" % (tag, scene)), "w").write("\n".join(lines) + "\n")

sys.path.insert(0, ROOT)
from moonshine_amd.hostinfo import source_hash      # noqa: E402  (ties the counters to the kernel sources they were taken on: bench.py flags them as stale otherwise)
out = {"command": open(G + "_cmd.txt").read().strip(), "source_hash": source_hash(), "steps": line.get("steps"), "unit": {"k_trace_closest": "closest-hit ray", "k_trace_shadow": "shadow ray", "k_shade": "path shaded (= closest-hit ray)"},
       "fetch_correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B; MI355X_MICROARCH.md HBM section), KB -> B x1024", "clock_ghz_assumed": 2.4, "kernels": {}}
per = {}
for sub in ("fetch", "write", "sq", "valu", "mem", "mem2"):
    if not os.path.exists(G + "_%s.log" % sub):
        continue
    tot = bench_line(G + "_%s.log" % sub)["profile_totals"]
    units = {"k_trace_closest": tot["closest_rays"], "k_trace_shadow": tot["shadow_rays"], "k_shade": tot["closest_rays"]}
    val, dur, n = counters(sub)
    per[sub] = (val, dur, n, units)
for k in KERNELS:
    fv, _, fn, fu = per["fetch"]; wv, _, _, wu = per["write"]; sv, sd, sn, su = per["sq"]
    e = {"dispatches": sn.get(k, 0), "units": su[k]}
    e["fetch_bytes_per_unit_raw"] = fv[k]["FETCH_SIZE"] * 1024.0 / fu[k]
    e["write_bytes_per_unit_raw"] = wv[k]["WRITE_SIZE"] * 1024.0 / wu[k]
    e["hbm_bytes_per_unit"] = 2.0 * e["fetch_bytes_per_unit_raw"] + e["write_bytes_per_unit_raw"]
    c = sv[k]
    e["valu_wave_instructions_per_unit"] = c["SQ_INSTS_VALU"] / su[k]
    e["valu_thread_instructions_per_unit"] = c["SQ_THREAD_CYCLES_VALU"] / su[k]
    e["lanes_per_valu_instruction"] = c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_INSTS_VALU"], 1.0)
    e["valu_issue_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * sd[k] * 2.4)
    e["kernel_ms_under_pmc"] = sd[k] * 1e-6
    e["vmem_rd_instructions_per_unit"] = c["SQ_INSTS_VMEM_RD"] / su[k]
    e["wave_cycles_split"] = {"active": c["SQ_ACTIVE_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0), "wait_inst": c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
                              "wait_any": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0)}
    if "valu" in per:   # instruction classes (tools/profile_round.sh pass "valu"): what bench.py's VALU roofline prices with profiles/r03_valu_calibration.json
        v, _, _, u = per["valu"]
        e["valu_class_per_unit"] = {c: v[k]["SQ_INSTS_VALU_" + c] / u[k] for c in ("FMA_F32", "MUL_F32", "ADD_F32", "INT32", "CVT", "TRANS_F32")}
        e["salu_instructions_per_unit"] = v[k]["SQ_INSTS_SALU"] / u[k]
    if "mem" in per:    # the vector-memory pipeline: TA / TD busy fractions (summed over the 256 CUs' units, per GRBM_GUI_ACTIVE cycle of one XCD), L1 tag accesses, L1 -> L2 requests
        v, d, _, u = per["mem"]
        cyc = v[k]["GRBM_GUI_ACTIVE"] / 8.0    # (the counter sums the 8 XCDs)
        e["memory_pipeline"] = {"ta_busy_frac": v[k]["TA_TA_BUSY_sum"] / 256.0 / cyc, "td_busy_frac": v[k]["TD_TD_BUSY_sum"] / 256.0 / cyc,
                                "tcp_pending_stall_frac": v[k]["TCP_PENDING_STALL_CYCLES_sum"] / 256.0 / cyc,
                                "l1_tag_accesses_per_unit": v[k]["TCP_TOTAL_CACHE_ACCESSES_sum"] / u[k], "l1_to_l2_read_requests_per_unit": v[k]["TCP_TCC_READ_REQ_sum"] / u[k],
                                "clock_ghz": cyc / max(d[k], 1.0)}
    if "mem2" in per:
        v, _, _, u = per["mem2"]
        e.setdefault("memory_pipeline", {}).update({"l2_hits_per_unit": v[k]["TCC_HIT_sum"] / u[k], "l2_misses_per_unit": v[k]["TCC_MISS_sum"] / u[k],
                                                     "vmem_read_instructions_per_unit": v[k]["SQ_INSTS_VMEM_RD"] / u[k]})
    out["kernels"][k] = e
json.dump(out, open(os.path.join(ROOT, "profiles", "%s_counters_%s.json" % (tag, scene)), "w"), indent=1)
print(json.dumps(out, indent=1))
