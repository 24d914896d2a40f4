"""Timeline of the LAST render in a rocprofv3 --kernel-trace csv: per-kernel time, overlap and idle gaps.
usage: python tools/timeline.py <kernel_trace.csv> [n_last_dispatches_of_k_shade=11]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 11
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("msne::", "")) for r in rows]
ev.sort()
shades = [i for i, e in enumerate(ev) if e[2].startswith("k_shade")]
first = shades[-n_last]
# the closest trace of bounce 0 and raygen come right before the first shade
while first > 0 and (ev[first - 1][2].startswith("k_trace_closest") or ev[first - 1][2].startswith("k_raygen")): first -= 1
sel = ev[first:]
t0 = sel[0][0]; t1 = max(e[1] for e in sel)
print("window %.3f ms, %d dispatches" % ((t1 - t0) / 1e6, len(sel)))
busy = 0; cur_s, cur_e = sel[0][0], sel[0][1]
for s, e, _ in sel[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("some kernel resident %.3f ms, idle gaps %.3f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
for s, e, n in sel: print("%9.3f %9.3f %8.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
