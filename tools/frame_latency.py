"""Wall time of ONE launch per render call (the interactive case: online / Hydra render one sample per pixel per frame): python tools/frame_latency.py [s1|cornell]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "s1"
c = api.Context()
s, l = (scenes.cornell(c, extent=(1920, 1080)) if which == "cornell" else scenes.s1(c, extent=(1920, 1080)))
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.reserve(s, 1)
for _ in range(5):
    c.render(s, l, launches=1, readback=False)
for rb in (False, True):
    t0 = time.perf_counter()
    for _ in range(100):
        c.render(s, l, launches=1, readback=rb)
    dt = (time.perf_counter() - t0) / 100
    print("%s 1920x1080, one launch per call, readback=%s: %.3f ms per frame (%.0f frames/s)" % (which, rb, dt * 1e3, 1 / dt))
