import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes
G = int(sys.argv[1]); L = int(sys.argv[2]) if len(sys.argv) > 2 else 64; T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
c = api.Context(tile_size=T, shard_index=0, shard_count=G)
s, l = scenes.s1(c, extent=(1920, 1080))
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.reserve(s, L)
c.render(s, l, launches=L, readback=False)
c.clear_sensor(s)
t0 = time.perf_counter(); c.render(s, l, launches=L, readback=False); print("ms", (time.perf_counter() - t0) * 1e3)
