"""Kernel timeline of the last batch in a rocprofv3 --kernel-trace CSV, with the rays of each trace launch when a bounce profile is given:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 tools/bounce_profile.py s1 64 8 > DIR/bounce.txt
    python tools/timeline.py DIR/p_kernel_trace.csv [DIR/bounce.txt]"""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "msne" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start = [i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]][-1]
paths, shadow = {}, {}
if len(sys.argv) > 2:
    for line in open(sys.argv[2]):
        m = re.match(r"bounce\s+(\d+): paths\s+(\d+) \(no ray\s+(\d+)\)\s+shadow entries\s+(\d+) traced\s+(\d+)", line)
        if m:
            b = int(m.group(1)); paths[b] = int(m.group(2)) - int(m.group(3)); shadow[b] = int(m.group(5))
t0 = int(rows[start]["Start_Timestamp"])
nb = {"k_trace_closest": 0, "k_trace_shadow": 1, "k_shade": 0}
prev_end = t0
for r in rows[start:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("msne::", "").split("<")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    extra = ""
    if name in nb:
        b = nb[name]; nb[name] += 1
        n = paths.get(b) if name != "k_trace_shadow" else shadow.get(b)
        if n:
            extra = "  bounce %2d  %10d units  %7.2f G/s" % (b, n, n / (e - s) / 1e6)
    print("%-18s %8.3f -> %8.3f  dur %7.3f%s" % (name, s, e, e - s, extra))
