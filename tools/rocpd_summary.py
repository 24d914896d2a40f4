"""Summarises a rocprofv3 (--kernel-trace --stats) rocpd database into a per-kernel table
(count, total/avg/min/max duration) — the form committed under profiles/.
    python tools/rocpd_summary.py gpurun_out/prof_r01/s1_results.db > profiles/r01_s1_kernel_stats.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info('kernels')")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary of %s" % sys.argv[1])
print("%-60s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for n, c, s, a, mn, mx in rows:
    short = n.split("(")[0]
    print("%-60s %8d %14d %12.0f %12d %12d %6.2f%%" % (short[:60], c, s, a, mn, mx, 100.0 * s / total))
