"""Host facts the measurement and the tests share."""
import os


def usable_cores():
    """hardware threads this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show 256 logical
    CPUs under a 16-CPU quota: 256 runnable threads there are throttled to 16 CPUs' worth of time, and os.cpu_count() says 256)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except Exception:
            pass
    return n


def source_hash():
    """sha1 over the kernel sources (csrc/*.hip, csrc/*.h) and the build flags: what per-ray instruction counters depend on.  tools/profile_counters.py records it
    in profiles/*_counters_*.json, bench.py compares it with the tree it runs from."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha1()
    d = os.path.join(here, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(here, "build.py"), "rb").read())
    return h.hexdigest()[:16]
