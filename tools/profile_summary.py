"""Turns the scratch output of tools/profile_round.sh TAG (gpurun_out/TAG_trace, TAG_fetch, TAG_write) into the files
committed under profiles/:  python tools/profile_summary.py TAG [timed_launches]

  profiles/TAG_kernel_stats.txt  rocprofv3 --kernel-trace --stats table of `python3 bench.py --no-cpu-baseline`, plus the
                                 per-kernel averages over the TIMED region only (the last `timed_launches` dispatches of each
                                 wavefront kernel; the earlier ones are bench.py's warm-up batch, which is 16x smaller)
  profiles/TAG_traffic.json      HBM bytes per timed k_trace_closest launch from the FETCH_SIZE / WRITE_SIZE passes, corrected
                                 as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2; both in KB)
"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
timed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
G = os.path.join(ROOT, "gpurun_out")


def short(n):
    return n.split("(")[0].replace("void ", "")


lines = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline   (MI355X, S1 1080p, 64 steps + 4 warm-up)",
         "%-40s %6s %14s %12s %10s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
for r in csv.DictReader(open(os.path.join(G, tag + "_trace", "p_kernel_stats.csv"))):
    lines.append("%-40s %6d %14d %12.0f %10d %12d %6.2f%%" % (short(r["Name"])[:40], int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"]),
                                                          int(r["MinNs"]), int(r["MaxNs"]), float(r["Percentage"])))
disp = defaultdict(list)
for r in csv.DictReader(open(os.path.join(G, tag + "_trace", "p_kernel_trace.csv"))):
    disp[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
lines.append("")
lines.append("# timed region only: the last %d dispatches of each wavefront kernel (bench.py: roofline.avg_launch_ms is measured over these)" % timed)
for k in sorted(disp):
    if not k.startswith("msne::k_trace") and k != "msne::k_shade":
        continue
    d = sorted(disp[k])[-timed:]
    tot = sum(e - s for s, e in d)
    lines.append("%-40s launches %3d  total %.3f ms  avg %.3f ms" % (k[:40], len(d), tot * 1e-6, tot * 1e-6 / len(d)))
open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")

out = {"workload": "bench.py default (S1 1920x1080, 64 steps, N=1)", "kernel": "k_trace_closest", "timed_launches": timed}
for name, sub in (("FETCH_SIZE", "_fetch"), ("WRITE_SIZE", "_write")):
    per = defaultdict(float)
    order = {}
    for r in csv.DictReader(open(os.path.join(G, tag + sub, "p_counter_collection.csv"))):
        if "k_trace_closest" in r["Kernel_Name"] and r["Counter_Name"] == name:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            order[r["Dispatch_Id"]] = int(r["Start_Timestamp"])
    last = sorted(per, key=lambda d: order[d])[-timed:]
    out[name + "_bytes_per_launch_raw"] = sum(per[d] for d in last) * 1024.0 / len(last)
out["fetch_correction"] = "x2 (MI355X_MICROARCH.md: gfx950 FETCH_SIZE tallies 128-B requests at 64 B; calibrated there for wide coalesced reads, so for 16-B-per-lane gathers it is an upper bound)"
out["traffic_bytes_per_launch"] = 2.0 * out["FETCH_SIZE_bytes_per_launch_raw"] + out["WRITE_SIZE_bytes_per_launch_raw"]
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_traffic.json"), "w"), indent=1)
print("\n".join(lines[-5:]))
print(json.dumps(out, indent=1))
