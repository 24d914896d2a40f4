"""The counters pass of tools/run_gather_microbench.sh as a table: per variant of tools/gather_microbench.hip (one dispatch each, W = 6, pools of 16 KiB and 2 MiB),
L1 tag accesses per clock per CU (TCP_TOTAL_CACHE_ACCESSES_sum / GRBM_GUI_ACTIVE / CUs), per lane-load, L1 -> L2 requests per lane-load, TA / TD busy.
    python tools/gather_counters.py DIR_OF_THE_PMC_RUN LOG_OF_THE_PMC_RUN"""
import csv, glob, os, re, sys
from collections import defaultdict, OrderedDict
d, log = sys.argv[1], sys.argv[2]
rows = OrderedDict()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        e = rows.setdefault(k, {"ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "grid": int(r.get("Grid_Size", 0) or 0)})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# the program prints its variants in dispatch order: pair them up
names = [l.split("  W=")[0].strip() + "  " + l.split("KiB")[0].split()[-1] + " KiB" for l in open(log) if " wave-fetches/us/CU" in l]
rates = [float(re.search(r"([\d.]+) wave-fetches/us/CU", l).group(1)) for l in open(log) if " wave-fetches/us/CU" in l]
CUS = 256
print("# one dispatch per variant under rocprofv3 --pmc (W = 6 waves per SIMD, 400 dependent hops per lane); tag = TCP_TOTAL_CACHE_ACCESSES_sum")
print("%-44s %9s %11s %12s %13s %8s %8s %9s" % ("variant", "fetch/us", "tag/clk/CU", "tag/laneload", "L2req/laneload", "TA busy", "TD busy", "clock GHz"))
disp = sorted(rows.items())
for i, ((did, kname), e) in enumerate(disp):
    m = re.search(r"k_gather<(\d+), (\d+), (\d+), (\d+)>", kname)
    nloads = int(m.group(1)) if m else 1
    laneloads = e["grid"] * 400.0 * nloads if e["grid"] else float("nan")
    act = e.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0      # (the counter sums the 8 XCDs: tools/profile_counters.py)
    tag = e.get("TCP_TOTAL_CACHE_ACCESSES_sum", float("nan"))
    print("%-44s %9.1f %11.3f %12.3f %13.3f %8.3f %8.3f %9.3f" % (names[i] if i < len(names) else kname[:44], rates[i] if i < len(rates) else float("nan"), tag / act / CUS, tag / laneloads,
          e.get("TCP_TCC_READ_REQ_sum", float("nan")) / laneloads, e.get("TA_TA_BUSY_sum", float("nan")) / act / CUS, e.get("TD_TD_BUSY_sum", float("nan")) / act / CUS, act / e["ns"]))
