"""Mrays/s of the other BASELINE.json configurations on one GPU: S1 with the sky+sun env (mip descent), S2 (10 M instanced
triangles: TLAS + transformed instances), Cornell 512x512.  64 launches each, wavefront state pre-allocated."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from moonshine_amd import api, scenes


def run(name, build, launches=64, **pipe):
    c = api.Context()
    s, l = build(c)
    c.set_pipeline(samples_per_run=1, max_bounces=8, **pipe)
    c.reserve(s, launches)
    c.set_profiling(True, False)
    c.render(s, l, launches=4, readback=False)
    c.clear_sensor(s); c.reset_stats()
    t0 = time.perf_counter(); c.render(s, l, launches=launches, readback=False); dt = time.perf_counter() - t0
    st = c.stats()
    rays = st["closest_rays"] + st["shadow_rays"]
    print("%-22s %8.1f Mrays/s %7.1f Msamples/s  %6.2f ms/launch  closest %.1f shadow %.1f shade %.1f ms" % (
        name, rays / dt / 1e6, st["samples"] / dt / 1e6, dt / launches * 1e3, st["trace_closest_ms"], st["trace_shadow_ms"], st["shade_ms"]))
    c.close()


run("S1 constant env", lambda c: scenes.s1(c), env_samples_per_bounce=1, mesh_samples_per_bounce=1)
run("S1 sky+sun env", lambda c: scenes.s1(c, env="sky"), env_samples_per_bounce=1, mesh_samples_per_bounce=1)
run("S2 10M instanced", lambda c: scenes.s2(c), env_samples_per_bounce=1, mesh_samples_per_bounce=1)
run("Cornell 512x512", lambda c: scenes.cornell(c), env_samples_per_bounce=0, mesh_samples_per_bounce=1)
