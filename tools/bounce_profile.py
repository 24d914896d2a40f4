"""Per-bounce queue lengths and kernel rates of one batch (S1 / S2 at 1080p): python tools/bounce_profile.py [s1|s2] [launches] [shards]
Run under MSNE_SERIAL=1 to time the kernels without co-residency."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "s1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c = api.Context(shard_index=0, shard_count=shards)
s, l = (scenes.s2 if scene == "s2" else scenes.s1)(c, extent=(1920, 1080))
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.reserve(s, K); c.set_profiling(True, False)
c.render(s, l, launches=4, readback=False)
c.reset_stats(); c.render(s, l, launches=K, readback=False)
st = c.stats()
print("%s x%d launches, shard 1/%d: render %.2f ms; closest %.2f shadow %.2f shade %.2f ms" % (scene, K, shards, st["render_ms"], st["trace_closest_ms"], st["trace_shadow_ms"], st["shade_ms"]))
for b, (n, z, sq, st_) in enumerate(c.bounce_counters(14)):
    print("bounce %2d: paths %10d (no ray %9d)  shadow entries %10d traced %10d" % (b, n, z, sq, st_))
