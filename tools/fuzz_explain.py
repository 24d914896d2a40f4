"""Why does a randomized scene of tests/test_gpu_parity.py::test_random_scenes_match_oracle fail?   gpurun -- python tools/fuzz_explain.py [--big] SEED [SEED ...]
Renders the seed's scene with the oracle and the HIP path, lists the film values that differ, and follows the oracle's camera paths of those pixels (every pixel when
only the ray counts differ) through OrcDebugPath: every ray of a path is traced again on the GPU (MsneTraceRays) and the first vertex whose hit differs is printed —
instance, geometry, primitive, t, u, v of both sides and the ray.  (Round 4: all five failures of seeds 0..20000 were coplanar triangles of two instances hit from
2e-3 away, tools/fuzz_sweep.sh.)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
np.set_printoptions(precision=9, linewidth=200)
from oracle import orc
from moonshine_amd import api
import test_gpu_parity as T
orc.build(); api.load_library()
BIG = "--big" in sys.argv      # seeds of test_random_big_scenes_match_oracle
for seed in [int(s) for s in sys.argv[1:] if s != "--big"]:
    if BIG:
        rs = np.random.default_rng(9000 + seed)
        oc, so, lo, gc, sg, lg = T.both(orc, api, T._random_scene, seed=70000 + seed, big=True)
        pipe = dict(samples_per_run=int(rs.integers(1, 4)), max_bounces=int(rs.integers(0, 9)), env_samples_per_bounce=int(rs.integers(0, 3)), mesh_samples_per_bounce=int(rs.integers(0, 3)), indexed_attributes=True, two_component_normal_texture=True)
        n = 2
    else:
        rs = np.random.default_rng(1000 + seed)
        oc, so, lo, gc, sg, lg = T.both(orc, api, T._random_scene, seed=seed)
        pipe = dict(samples_per_run=int(rs.integers(1, 3)), max_bounces=int(rs.integers(0, 7)), env_samples_per_bounce=int(rs.integers(0, 3)), mesh_samples_per_bounce=int(rs.integers(0, 3)), indexed_attributes=True, two_component_normal_texture=True)
        n = int(rs.integers(1, 4))
    print("seed", seed, pipe, "launches", n)
    for c in (oc, gc):
        c.set_pipeline(**pipe)
    gc.render(sg, lg, launches=n); oc.render(so, lo, launches=n)
    go, oo = gc.sensor_data(sg), oc.sensor_data(so)
    same = (go.view(np.uint32) == oo.view(np.uint32)) | (np.isnan(go) & np.isnan(oo))
    H, W = go.shape[:2]
    px = sorted(set((int(y), int(x)) for y, x, _ in np.argwhere(~same)))
    print("  film %dx%d differs at" % (W, H), px)
    spp = pipe["samples_per_run"] * n
    pixels = px if px else [(y, x) for y in range(H) for x in range(W)]
    tot_c = tot_s = 0
    for (y, x) in pixels:
        for k in range(spp):
            rgb, rec, cnt = oc.debug_path(so, lo, k, x, y)
            tot_c += cnt[0]; tot_s += cnt[1]
            if not len(rec): continue
            rays = np.concatenate([rec[:, 8:11], rec[:, 5:8], np.full((len(rec), 1), np.inf, np.float32)], axis=1).astype(np.float32)
            ids, tuv = gc.trace_rays(rays)
            for i in range(len(rec)):
                o_ids = (int(rec[i, 0]), int(rec[i, 11]), int(rec[i, 1]))
                if not ids[i, 0] or tuple(int(v) for v in ids[i, 1:4]) != o_ids or not np.array_equal(tuv[i].view(np.uint32), rec[i, 2:5].view(np.uint32)):
                    print("  px", (x, y), "k", k, "vertex", i, "orc", o_ids, rec[i, 2:5], "gpu", ids[i], tuv[i], "ray", rays[i])
            if px:
                print("  px", (x, y), "k", k, "orc rgb", rgb, "hits", len(rec), "counts", cnt)
    if px:
        for (y, x) in px: print("  film gpu", go[y, x], "orc", oo[y, x])
    else:
        print("  oracle per-path totals", tot_c, tot_s, "render", oc.counters()["closest_rays"], oc.counters()["shadow_rays"])
