"""Kernel times of S1 1080p x 64 launches under different NEE settings (which part of k_shade costs what)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes

c = api.Context()
s, l = scenes.s1(c, extent=(1920, 1080))
for env_n, mesh_n in ((1, 1), (0, 1), (1, 0), (0, 0)):
    c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=env_n, mesh_samples_per_bounce=mesh_n)
    c.reserve(s, 64)
    c.set_profiling(True, False)
    c.render(s, l, launches=4, readback=False)
    c.reset_stats()
    c.render(s, l, launches=64, readback=False)
    st = c.stats()
    print("env=%d mesh=%d closest_rays=%d shadow_rays=%d closest=%.1f shadow=%.1f shade=%.1f render=%.1f ms" % (
        env_n, mesh_n, st["closest_rays"], st["shadow_rays"], st["trace_closest_ms"], st["trace_shadow_ms"], st["shade_ms"], st["render_ms"]))
