"""usage (GPU box): python tools/hull_ray_explain.py SEED [SEED ...] — tests/test_gpu_parity.py::test_rays_at_the_hulls_of_far_scaled_and_sheared_instances ray by ray: every
ray whose hit record or occlusion differs between the HIP path and the oracle, with both records, the oracle's three search levels (BVH / every instance entered / every
triangle tested), the instance's transform and the hit triangle in instance space."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
np.set_printoptions(precision=9, linewidth=220)
from oracle import orc
import moonshine_amd.api as api
import hull_rays
orc.build()
for seed in [int(a) for a in sys.argv[1:] if not a.startswith('--')]:
    oc = orc.Context(threads=8); gc = api.Context()
    harsh = seed % 2 == 1; baked = seed % 3 == 2
    parts = []
    world = hull_rays.hull_scene(oc, seed, harsh, parts, baked); hull_rays.hull_scene(gc, seed, harsh, baked=baked)
    for c in (oc, gc): c.create_sensor(8, 8)
    def check(rays, tag):
        ids, tuv = gc.trace_rays(rays, any_hit=False); occ, _ = gc.trace_rays(rays, any_hit=True)
        for k, r in enumerate(rays):
            lv = []
            for level in (0, 1, 2):
                oc.set_exhaustive_search(level); lv.append((oc.trace_closest(r[:3], r[3:6], float(r[6])), oc.trace_shadow(r[:3], r[3:6], float(r[6]))))
            oc.set_exhaustive_search(0)
            (hit, oid, otuv), osh = lv[0]
            bad = bool(ids[k, 0]) != hit or (hit and (tuple(ids[k, 1:4]) != tuple(oid) or not np.array_equal(tuv[k].view(np.uint32), otuv.view(np.uint32)))) or bool(occ[k, 0]) != osh
            if bad:
                print(seed, tag, "ray", k, [float(x).hex() for x in r], "\n   ", r)
                print("    hip closest", ids[k], tuv[k], "occluded", bool(occ[k, 0]))
                for level, (cl, sh) in zip((0, 1, 2), lv): print("    oracle level", level, cl, "occluded", sh)
        print(seed, tag, "checked", len(rays), "rays; accel", gc.accel_stats(), flush=True)
    check(hull_rays.hull_rays(world, seed), "first")
    if baked: continue
    hull_rays.hull_move((oc, gc), seed, parts, world)
    check(hull_rays.hull_rays(world, seed + 1)[::2], "moved")
    if "--rebuild" in sys.argv:      # the same rays once more on a TLAS that is BUILT for the moved instances (a visibility edit rebuilds): is it the re-fit?
        for c in (oc, gc): c.set_instance_visibility(1, False)
        gc.trace_rays(hull_rays.hull_rays(world, seed + 1)[:2])
        for c in (oc, gc): c.set_instance_visibility(1, True)
        check(hull_rays.hull_rays(world, seed + 1)[::2], "moved, rebuilt")
    hull_rays.hull_move((oc, gc), seed + 5, parts, world)
    check(hull_rays.hull_rays(world, seed + 2, far=1.0)[::3], "moved2")
