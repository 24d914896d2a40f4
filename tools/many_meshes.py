"""Scalability probe: N DISTINCT small meshes, each its own instance with a transform (N BLAS builds + one TLAS): time of the first render (build) and of a
re-render.  python tools/many_meshes.py [N=2000] [order=2]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from moonshine_amd import api, scenes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
order = int(sys.argv[2]) if len(sys.argv) > 2 else 2
c = api.Context()
P, I = scenes.icosphere(order)
rs = np.random.default_rng(0)
t0 = time.perf_counter()
mat = c.create_material(scenes.LAMBERT, c.solid_texture(0.5, 0.5), c.solid_texture(0, 0, 0), color=c.solid_texture(0.7, 0.7, 0.7))
side = int(np.ceil(N ** (1 / 3)))
for k in range(N):
    m = c.create_mesh((P * rs.uniform(0.6, 1.0, (1, 3))).astype(np.float32), I)     # distinct vertex data: no BLAS is shared
    T = np.zeros((3, 4), np.float32); T[:, :3] = np.eye(3) * 0.4; T[:, 3] = (k % side, (k // side) % side, k // (side * side))
    T[0, 1] = 0.05                                                                   # not an identity: stays out of the world BLAS
    c.create_instance([(m, mat, False)], transform=T)
s = c.create_sensor(640, 360); l = c.create_lens(c.make_lens((-side, side / 2, side / 2), (1, 0, 0), (0, 0, 1), 0.9))
c.set_pipeline(samples_per_run=1, max_bounces=4, env_samples_per_bounce=1, mesh_samples_per_bounce=0)
t1 = time.perf_counter()
c.render(s, l, launches=1)
t2 = time.perf_counter()
c.render(s, l, launches=1)
t3 = time.perf_counter()
print("%d meshes x %d triangles: scene calls %.2f s, first render (BLAS + TLAS build) %.3f s = %.3f ms per mesh, second render %.4f s" % (N, len(I), t1 - t0, t2 - t1, (t2 - t1) / N * 1e3, t3 - t2))
