"""What k_shade waits for: the three truncated instantiations (integrator.hip TRUNC = 1 path-state loads + LDS staging, 2 = + hit / geometry / material chain + sort,
3 = + attributes, frames, material parameters) run in front of the real kernel on the same queues ($MSNE_SHADE_TRUNC=1), under rocprofv3.
  GPU box:  MSNE_SHADE_TRUNC=1 bash tools/profile_round.sh r05t s1 trace,mem,mem2,sq --steps 20
  here:     python tools/shade_td.py r05t s1  > profiles/r05_shade_td.txt
Per instantiation: dispatches, total and mean duration, and per dispatch-microsecond the memory-pipeline counters (TA / TD busy, L1 tag accesses, L1->L2 requests,
L2 hits / misses, pending-stall cycles) and the wave-cycle split."""
import csv, glob, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, scene = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out", "%s_%s" % (tag, scene))


def name_of(k):
    m = re.search(r"k_shade<(\d), (true|false)(?:, (\d))?>", k)
    if not m:
        return None
    return "k_shade" + ("<TRUNC %s>" % m.group(3) if m.group(3) and m.group(3) != "0" else " (full)")


def counters(sub):
    val = defaultdict(lambda: defaultdict(float)); dur = defaultdict(float); seen = defaultdict(set)
    for f in glob.glob(os.path.join(G + "_" + sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = name_of(r["Kernel_Name"])
            if not k:
                continue
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen[k]:
                seen[k].add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return val, dur, {k: len(v) for k, v in seen.items()}


trace = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(G + "_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = name_of(r["Kernel_Name"])
        if k:
            trace[k][0] += 1; trace[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
order = ["k_shade<TRUNC %d>" % i for i in (1, 2, 3, 7, 4, 5, 6)] + ["k_shade (full)"]
print("# %s — k_shade cut after each of its stages (tools/shade_td.py %s %s; S1, the driver's 20 steps; the first pass of a batch — camera rays, no state to load — is not truncated)" % (open(G + "_cmd.txt").read().strip(), tag, scene))
print("# kernel trace (no counters): dispatches, total ms, mean us per dispatch, and what each stage adds")
prev = 0.0
for k in order:
    if k in trace:
        n, ns = trace[k]
        print("  %-18s %4d dispatches  %8.3f ms  %8.1f us each   (+%.3f ms over the stage before)" % (k, n, ns / 1e6, ns / 1e3 / n, ns / 1e6 - prev)); prev = ns / 1e6
for sub, title in (("mem", "memory pipeline, counters per microsecond of the kernel's own dispatches (TA/TD busy: fraction of the sum over the chip's 256 TA/TD instances)"),
                   ("mem2", "L2"), ("sq", "wave cycles")):
    val, dur, nd = counters(sub)
    if not val:
        continue
    print("# pass %s — %s" % (sub, title))
    for k in order:
        if k not in val:
            continue
        v = val[k]; us = dur[k] / 1e3
        cyc = v.get("GRBM_GUI_ACTIVE", 0.0)
        parts = []
        for c in sorted(v):
            if c in ("TA_TA_BUSY_sum", "TD_TD_BUSY_sum", "TCP_PENDING_STALL_CYCLES_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum") and cyc:
                parts.append("%s %.3f of 256 x GUI_ACTIVE" % (c.replace("_sum", ""), v[c] / (256.0 * cyc)))
            elif c.startswith("SQ_WAIT") or c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") :
                parts.append("%s %.3f of SQ_WAVE_CYCLES" % (c, v[c] / max(v.get("SQ_WAVE_CYCLES", 1.0), 1.0)))
            else:
                parts.append("%s %.1f /us" % (c.replace("_sum", ""), v[c] / max(us, 1e-9)))
        print("  %-18s %4d dispatches %8.3f ms | %s" % (k, nd[k], dur[k] / 1e6, " | ".join(parts)))
