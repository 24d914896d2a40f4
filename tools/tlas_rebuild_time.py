import sys, time, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/moonshine_amd') else '.')
import torch
from moonshine_amd import api, scenes
import numpy as np
for name, kw in (("S2 500 x 10242 verts", dict(extent=(320, 180), dims=(10, 10, 5), order=5)), ("100k x 42 verts", dict(extent=(320, 180), dims=(50, 50, 40), order=1))):
    c = api.Context()
    s, l = scenes.s2(c, **kw)
    c.set_pipeline(samples_per_run=1, max_bounces=2, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    t0 = time.perf_counter(); c.render(s, l); t1 = time.perf_counter()
    T = np.eye(3, 4, dtype=np.float32); T[:, 3] = [0.1, 0.2, 0.3]
    c.set_instance_transform(3, T)
    t2 = time.perf_counter(); c.render(s, l); t3 = time.perf_counter()
    print("%s  exact=%s: first render (BLAS+TLAS build) %.1f ms, render after one instance edit (TLAS rebuild) %.1f ms" % (name, os.environ.get("MSNE_EXACT_INSTANCE_BOXES", "1"), (t1 - t0) * 1e3, (t3 - t2) * 1e3))
    c.close()
