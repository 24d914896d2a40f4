"""L2 hit rate and duration of every traversal launch of a batch, in launch order (= by bounce), from a rocprofv3 PMC run:
    (GPU box) cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/DIR -o p -- python3 $R/tools/shard_one.py 8 20 16
    (here)    python tools/l2_by_bounce.py gpurun_out/DIR [launches-per-batch-of-each-kernel]
Rows: the LAST batch's dispatches of k_trace_closest / k_trace_shadow / k_trace_probe.  TCC_HIT / TCC_MISS are summed over the 16 channels x 8 XCDs (128-B requests)."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
disp = defaultdict(lambda: {"name": "", "t0": 0, "t1": 0, "c": defaultdict(float)})
for r in csv.DictReader(open(f)):
    e = disp[int(r["Dispatch_Id"])]
    e["name"] = r["Kernel_Name"]; e["t0"] = int(r["Start_Timestamp"]); e["t1"] = int(r["End_Timestamp"]); e["grid"] = int(r["Grid_Size"]) if "Grid_Size" in r else 0
    e["c"][r["Counter_Name"]] += float(r["Counter_Value"])
for kname in ("k_trace_closest", "k_trace_shadow", "k_trace_probe"):
    rows = [e for _, e in sorted(disp.items()) if "msne::" + kname in e["name"]]
    if not rows:
        continue
    per = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
    rows = rows[-per:]
    print("%s: the last %d dispatches" % (kname, len(rows)))
    for i, e in enumerate(rows):
        h, m = e["c"].get("TCC_HIT_sum", 0.0), e["c"].get("TCC_MISS_sum", 0.0)
        print("  launch %2d  %8.1f us   L2 requests %10.0f   hits %10.0f   misses %10.0f   hit rate %.3f" % (i, (e["t1"] - e["t0"]) / 1e3, h + m, h, m, h / max(h + m, 1.0)))
