"""A/B timing of library variants on one GPU:   python tools/variant_rates.py [--scenes s1,s1k20,s2,sky,cornell] VARIANT [VARIANT ...]
VARIANT = "default" or the name of a build made with moonshine_amd.build.build(variant=NAME, extra_flags=[...]) (libmoonshine_amd_NAME.so).
Every (variant, scene) runs in its own process (MSNE_LIB selects the library), kernels in stream order (MSNE_SERIAL=1) so the per-kernel
times are clean; the rate is the median of 3 batches."""
import os, subprocess, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch  # noqa
    from moonshine_amd import api, scenes
    scene = sys.argv[2]
    launches = 20 if scene.endswith("k20") else 64
    c = api.Context()
    if scene.startswith("s1"): s, l = scenes.s1(c)
    elif scene == "sky": s, l = scenes.s1(c, env="sky")
    elif scene == "s2": s, l = scenes.s2(c)
    else: s, l = scenes.cornell(c)
    c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=0 if scene == "cornell" else 1, mesh_samples_per_bounce=1)
    c.reserve(s, launches); c.set_profiling(True, False)
    c.render(s, l, launches=4, readback=False)
    rates, ks = [], None
    for _ in range(3):
        c.clear_sensor(s); c.reset_stats()
        t0 = time.perf_counter(); c.render(s, l, launches=launches, readback=False); dt = time.perf_counter() - t0
        st = c.stats(); rates.append((st["closest_rays"] + st["shadow_rays"]) / dt / 1e6)
        ks = (st["trace_closest_ms"], st["trace_shadow_ms"], st["shade_ms"], dt * 1e3)
    print("%-10s %-8s %8.1f Mrays/s (%.1f..%.1f)  closest %6.2f shadow %6.2f shade %6.2f total %7.2f ms" % (
        os.environ.get("MSNE_VARIANT", "?"), scene, statistics.median(rates), min(rates), max(rates), *ks), flush=True)
    sys.exit(0)

scenes_ = ["s1", "s1k20", "s2"]
args = sys.argv[1:]
if args and args[0] == "--scenes": scenes_ = args[1].split(","); args = args[2:]
for sc in scenes_:
    for v in args:
        env = dict(os.environ, MSNE_VARIANT=v, MSNE_SERIAL=os.environ.get("MSNE_SERIAL", "1"))
        if v != "default": env["MSNE_LIB"] = os.path.join(ROOT, "moonshine_amd", "libmoonshine_amd_%s.so" % v)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", sc], env=env)
