"""Sums rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py gpurun_out/<dir> [...]
Prints one line per (pass, kernel): dispatches, total duration (ms) and every counter summed over the dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict


def summarise(d):
    rows = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if not k.startswith("msne::") and "msne" not in k:
                continue
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in disp[k]:
                disp[k].add(r["Dispatch_Id"])
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for k in sorted(rows):
        print("%s %-34s n=%d ms=%.2f %s" % (os.path.basename(d.rstrip("/")), k, len(disp[k]), dur[k],
                                           " ".join("%s=%.4g" % kv for kv in sorted(rows[k].items()))))


if __name__ == "__main__":
    for d in sys.argv[1:]:
        summarise(d)
