"""Measures the per-scene constants of SURVEY.md §8(d)'s roofline formula with the CPU oracle's canonical
BVH (LBVH, 30-bit Morton, leaves <= 4, collapsed to <= 8-wide, ordered traversal) and writes
tests/golden/roofline_<scene>.json:  V_n / V_t = mean BVH-node visits / triangle tests per ray over the run's
full ray population (closest + shadow), H = mean surface hits per sample.

    python tools/make_roofline_fixture.py s1 --width 1920 --height 1080 --spp 64      (the committed fixtures: the full ray population of the bench run)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc          # noqa: E402  (tools/ is test infrastructure: it produces golden fixtures)
from moonshine_amd import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scene", choices=["s1", "s1_sky", "s2", "cornell", "standin"])
ap.add_argument("--width", type=int, default=480)
ap.add_argument("--height", type=int, default=270)
ap.add_argument("--spp", type=int, default=4)
a = ap.parse_args()

orc.build()
from moonshine_amd.hostinfo import usable_cores  # noqa: E402
c = orc.Context(threads=usable_cores())
t0 = time.time()
if a.scene == "s1":
    s, l = scenes.s1(c, extent=(a.width, a.height))
elif a.scene == "s1_sky":
    s, l = scenes.s1(c, extent=(a.width, a.height), env="sky")
elif a.scene == "s2":
    s, l = scenes.s2(c, extent=(a.width, a.height))
elif a.scene == "standin":      # configs[2]'s stand-in (tests/io_common.py write_bathroom_standin), loaded the way the parity tests load it
    import tempfile
    from tests import io_common as io
    d = tempfile.mkdtemp(); glb, exr = os.path.join(d, "bath.glb"), os.path.join(d, "sky.exr")
    io.write_bathroom_standin(glb, exr)
    l, _ = io.oracle_load(orc, c, glb, exr); s = c.create_sensor(a.width, a.height)
else:
    s, l = scenes.cornell(c, extent=(a.width, a.height))
nee = (0, 1) if a.scene == "cornell" else (1, 1)
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=nee[0], mesh_samples_per_bounce=nee[1])
c.render(s, l, launches=a.spp)
k = c.counters()
rays = k["closest_rays"] + k["shadow_rays"]
out = {
    "scene": a.scene, "extent": [a.width, a.height], "spp": a.spp, "max_bounces": 8, "nee": list(nee),
    "closest_rays": k["closest_rays"], "shadow_rays": k["shadow_rays"], "samples": k["samples"], "surface_hits": k["surface_hits"],
    "V_n": k["node_visits"] / rays, "V_t": k["tri_tests"] / rays,
    "V_n_closest": (k["node_visits"] - k["shadow_node_visits"]) / k["closest_rays"],
    "V_t_closest": (k["tri_tests"] - k["shadow_tri_tests"]) / k["closest_rays"],
    "V_n_shadow": k["shadow_node_visits"] / max(k["shadow_rays"], 1), "V_t_shadow": k["shadow_tri_tests"] / max(k["shadow_rays"], 1),
    "H": k["surface_hits"] / k["samples"], "rays_per_sample": rays / k["samples"],
    "B_hit": 244,
}
out["B_ray"] = out["V_n"] * 80 + out["V_t"] * 48 + 48
out["B_shade"] = out["B_hit"] * out["H"]
p = os.path.join(ROOT, "tests", "golden", "roofline_%s.json" % a.scene)
json.dump(out, open(p, "w"), indent=1)
print(json.dumps(out, indent=1))
print("wrote", p, "in %.1fs" % (time.time() - t0))
