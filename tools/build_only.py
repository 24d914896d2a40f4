"""Builds one scene's acceleration structures and exits: the subject of builder profiles (rocprofv3 --kernel-trace --stats -- python3 tools/build_only.py s1).
usage: python tools/build_only.py [s1|s2|big|standin] [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "s1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for _ in range(reps):
    c = api.Context()
    t0 = time.perf_counter()
    if which == "s2":
        s, l = scenes.s2(c, extent=(64, 36))
    elif which == "big":
        s, l = scenes.s1(c, extent=(64, 36), grid=28, order=5)
    else:
        s, l = scenes.s1(c, extent=(64, 36))
    t1 = time.perf_counter()
    c.set_pipeline(samples_per_run=1, max_bounces=1, env_samples_per_bounce=0, mesh_samples_per_bounce=0)
    c.render(s, l, launches=1, readback=False)
    t2 = time.perf_counter()
    print("%s: scene upload %.1f ms, build + one 64x36 launch %.1f ms" % (which, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
    c.close()
