"""GPU-side traversal counters on S1 (node visits / triangle tests per ray) next to the oracle fixture's V_n/V_t."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes
w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 270, 4)
c = api.Context(shard_index=0, shard_count=int(os.environ.get("SHARDS", "1")))   # SHARDS=8: rank 0 of an 8-way tile shard
s, l = (scenes.s2 if os.environ.get("SCENE") == "s2" else scenes.s1)(c, extent=(w, h))
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.set_profiling(True, True)
c.render(s, l, launches=spp, readback=False)
st = c.stats(); t = c.traversal_counters()
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "roofline_s1.json")))
print(json.dumps({"closest_rays": st["closest_rays"], "shadow_rays": st["shadow_rays"],
                  "gpu_V_n_closest": t["closest_node_visits"] / st["closest_rays"], "gpu_V_t_closest": t["closest_tri_tests"] / st["closest_rays"],
                  "gpu_V_n_shadow": t["shadow_node_visits"] / st["shadow_rays"], "gpu_V_t_shadow": t["shadow_tri_tests"] / st["shadow_rays"],
                  "oracle": {k: fx[k] for k in ("V_n_closest", "V_t_closest", "V_n_shadow", "V_t_shadow")},
                  "closest_profile": t["closest_profile"], "shadow_profile": t["shadow_profile"],
                  "ms": {k: st[k] for k in ("trace_closest_ms", "trace_shadow_ms", "shade_ms", "render_ms")}}))
