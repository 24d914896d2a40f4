"""Lane use of the traversal loop per wave iteration (STATS instantiation of the trace kernels): who is in which body, who waits for what.
usage: SCENE=s2 python tools/lane_use.py [w h spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes
w, h, spp = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080, 4)
c = api.Context()
name = os.environ.get("SCENE", "s2")
s, l = (scenes.s2 if name == "s2" else scenes.s1)(c, extent=(w, h))
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.set_profiling(True, True)
c.render(s, l, launches=spp, readback=False)
st = c.stats(); t = c.traversal_counters(); u = c.traversal_lane_use()
for k, rays, nv, nt in (("closest", st["closest_rays"], t["closest_node_visits"], t["closest_tri_tests"]), ("shadow", st["shadow_rays"], t["shadow_node_visits"], t["shadow_tri_tests"])):
    x = u[k]; it = max(x["iterations"], 1)
    print("%s k_trace_%s: %d rays, %.2f node visits, %.2f triangle tests, %.2f space changes per ray; %.1f wave iterations per 64 rays" % (name.upper(), k, rays, nv / rays, nt / rays, x["space_body"] / rays, it * 64.0 / rays))
    print("  per iteration (lanes of 64): with a ray %.1f | node body %.1f (runs in %.0f %% of the iterations, %.1f lanes when it runs) | triangle body %.1f (%.0f %%, %.1f) | space body %.1f (%.0f %%, %.1f)"
          % (x["with_ray"] / it, x["node_body"] / it, 100.0 * x["iter_node"] / it, x["node_body"] / max(x["iter_node"], 1), x["tri_body"] / it, 100.0 * x["iter_tri"] / it, x["tri_body"] / max(x["iter_tri"], 1),
             x["space_body"] / it, 100.0 * x["iter_space"] / it, x["space_body"] / max(x["iter_space"], 1)))
    print("  waiting: for the space body %.1f | for the triangle queue to drain %.1f | with a ray but in no body %.1f | without a ray %.1f" % (x["wait_space"] / it, x["wait_tri_queue"] / it, x["no_body"] / it, 64.0 - x["with_ray"] / it))
