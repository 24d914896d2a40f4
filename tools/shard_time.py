"""Per-rank time of an N-way sharded S1 run measured on ONE GPU (rank 0's tiles only) — predicts strong scaling."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes
out = {}
for G in (1, 2, 4, 8):
    c = api.Context(shard_index=0, shard_count=G)
    s, l = scenes.s1(c, extent=(1920, 1080))
    c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    c.reserve(s, 64)
    c.render(s, l, launches=4, readback=False)
    c.clear_sensor(s); c.reset_stats()
    t0 = time.perf_counter(); c.render(s, l, launches=64, readback=False); dt = time.perf_counter() - t0
    st = c.stats()
    out[G] = {"ms": dt * 1e3, "rays": st["closest_rays"] + st["shadow_rays"]}
    c.close()
for G in out:
    print("shards=%d rank0 time %.2f ms  predicted speedup %.2fx (efficiency %.0f%%)" % (G, out[G]["ms"], out[1]["ms"] / out[G]["ms"], 100 * out[1]["ms"] / out[G]["ms"] / G))
