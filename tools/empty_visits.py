"""Node visits none of whose children the ray hits (a library built with -DTRACE_COUNT_EMPTY=1: all of them; =2: those made with a closest hit already found; =3: visits whose node lies, as a whole, beyond
the limit the visit tests against — what a bound carried in the stack entry could cull when the entry is popped), per ray, S1 and S2 at 480x270:
    MSNE_LIB=moonshine_amd/libmoonshine_amd_cnt.so python tools/empty_visits.py   (profiles/r05_tri_density.txt section 8)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa
from moonshine_amd import api, scenes
for name in ("s1", "s2"):
    c = api.Context()
    s, l = (scenes.s2 if name == "s2" else scenes.s1)(c, extent=(480, 270))
    c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    c.set_profiling(True, True)
    c.render(s, l, launches=1, readback=False)
    st = c.stats(); t = c.traversal_counters()
    for k in ("closest", "shadow"):
        v = t[k + "_node_visits"]; rays = st[k + "_rays"]
        print(os.environ.get("MSNE_LIB", "")[-8:], name, k, "node visits per ray %.2f, counted %.2f" % ((v & 0xffffffff) / rays, (v >> 32) / rays))
