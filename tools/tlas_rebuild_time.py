"""What an instance edit costs: first render (BLAS + TLAS build), a render after ONE transform edit (in-place TLAS update, Accel.zig:567-601), a render
after an edit that changes the structure (visibility: TLAS rebuild), and a plain render for reference — S2 (500 instances) and 100 000 small instances."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from moonshine_amd import api, scenes
import numpy as np


def timed(f):
    t0 = time.perf_counter(); f(); return (time.perf_counter() - t0) * 1e3


for name, kw in (("S2 500 x 10242 verts", dict(extent=(320, 180), dims=(10, 10, 5), order=5)), ("100k x 42 verts", dict(extent=(320, 180), dims=(50, 50, 40), order=1))):
    c = api.Context()
    s, l = scenes.s2(c, **kw)
    c.set_pipeline(samples_per_run=1, max_bounces=2, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    first = timed(lambda: c.render(s, l))
    plain = min(timed(lambda: c.render(s, l)) for _ in range(3))
    T = np.zeros((3, 4), dtype=np.float32); T[:, :3] = np.eye(3) * 0.9; T[:, 3] = [0.1, 0.2, 3.3]
    c.set_instance_transform(3, T)
    upd = timed(lambda: c.render(s, l))
    rs = np.random.default_rng(1); n_inst = kw["dims"][0] * kw["dims"][1] * kw["dims"][2]
    many = rs.choice(n_inst, n_inst // 10, replace=False)
    dx, dy, dz = kw["dims"]
    for h in many:     # every moved instance stays near its own grid cell (scenes.s2's layout): the scene keeps its character, only the TLAS changes
        ix, iy, iz = int(h) % dx, (int(h) // dx) % dy, int(h) // (dx * dy)
        T2 = T.copy(); T2[:, 3] = np.array([(ix - (dx - 1) / 2) * 2.5, (iy - (dy - 1) / 2) * 2.5, 1.0 + iz * 2.5]) + rs.normal(size=3) * 0.5
        c.set_instance_transform(int(h), T2)
    upd_many = timed(lambda: c.render(s, l))
    many_stats = c.accel_stats()
    c.set_instance_visibility(5, False)
    reb = timed(lambda: c.render(s, l))
    print("%-22s first render (BLAS + TLAS build) %6.1f ms | plain render %5.1f ms | after one transform edit %5.1f ms (%s) | after %d transform edits at once %5.1f ms (%s) | after a visibility edit (TLAS rebuild) %5.1f ms"
          % (name, first, plain, upd, c.accel_stats(), len(many), upd_many, many_stats, reb))
    c.close()
