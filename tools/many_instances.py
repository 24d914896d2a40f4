"""Scalability probe: 100 000 instances of one small mesh (TLAS build + traversal).  python tools/many_instances.py"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moonshine_amd import api, scenes
c = api.Context()
t0 = time.perf_counter()
s, l = scenes.s2(c, extent=(640, 360), dims=(50, 50, 40), order=1)
c.set_pipeline(samples_per_run=1, max_bounces=4, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
t1 = time.perf_counter()
c.render(s, l, launches=1)
t2 = time.perf_counter()
c.render(s, l, launches=8)
t3 = time.perf_counter()
img = c.sensor_data(s)
print("100k instances: create %.2f s, first render %.2f s, 8 launches %.1f ms, finite %s, mean %.4f, stats %s" % (t1 - t0, t2 - t1, (t3 - t2) * 1e3, np.isfinite(img).all(), img[..., :3].mean(), c.counters()))
