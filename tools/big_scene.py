"""Scalability probe: S1's layout with grid x grid icospheres (grid=28, order=5: 16 M triangles in ONE BLAS).
usage: python tools/big_scene.py [grid=28] [order=5]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 28
order = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = api.Context()
t0 = time.perf_counter()
s, l = scenes.s1(c, extent=(1920, 1080), grid=grid, order=order)
t1 = time.perf_counter()
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.render(s, l, launches=1, readback=False)     # includes the BVH build
t2 = time.perf_counter()
c.reset_stats(); c.clear_sensor(s)
c.render(s, l, launches=16, readback=False)
t3 = time.perf_counter()
st = c.stats()
rays = st["closest_rays"] + st["shadow_rays"]
print("grid %d order %d: %d triangles; scene upload %.2f s, first launch incl. BVH build %.3f s, 16 launches %.1f ms = %.0f Mrays/s"
      % (grid, order, grid * grid * 20 * 4 ** order, t1 - t0, t2 - t1, (t3 - t2) * 1e3, rays / (t3 - t2) / 1e6))
img = c.sensor_data(s) if hasattr(c, "sensor_data") else None
