"""Per-rank time of an N-way sharded S1 run measured on ONE GPU, one rank's tiles at a time — predicts strong scaling.

The job time at G shards is the MAX over its ranks (bench.py takes the max over ranks too), so every rank of every G is
timed, not only rank 0.  `python tools/shard_time.py [--quick] [--tile N] [--steps K]` (--quick: rank 0 only; --steps: launches per batch, 64 by
default; 20 = the command the driver's scaling run uses)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes

quick = "--quick" in sys.argv
tile = int(sys.argv[sys.argv.index("--tile") + 1]) if "--tile" in sys.argv else 0   # 0 = the library default (16)
K = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 64
out = {}
for G in (1, 2, 4, 8):
    per_rank = []
    for r in range(1 if quick else G):
        c = api.Context(tile_size=tile, shard_index=r, shard_count=G)
        s, l = scenes.s1(c, extent=(1920, 1080))
        c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
        c.reserve(s, K)
        c.render(s, l, launches=4, readback=False)
        best = 1e30
        for _ in range(3):
            c.clear_sensor(s); c.reset_stats()
            t0 = time.perf_counter(); c.render(s, l, launches=K, readback=False); best = min(best, time.perf_counter() - t0)
        st = c.stats()
        per_rank.append((best * 1e3, st["closest_rays"] + st["shadow_rays"]))
        c.close()
    out[G] = per_rank
t1 = out[1][0][0]
for G, pr in out.items():
    ms = [p[0] for p in pr]; rays = [p[1] for p in pr]
    worst = max(ms)
    print("steps=%d tile=%d shards=%d  rank ms min %.2f max %.2f  rays/rank min %.1fM max %.1fM  predicted speedup %.2fx (efficiency %.0f%%)"
          % (K, tile, G, min(ms), worst, min(rays) / 1e6, max(rays) / 1e6, t1 / worst, 100 * t1 / worst / G))
