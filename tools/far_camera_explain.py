"""usage (GPU box): python tools/far_camera_explain.py SEED FAR — tests/test_gpu_parity.py::test_camera_far_outside_the_baked_reach's film of one (seed, far): the pixels
that differ between the HIP path and the oracle's search without boxes, and for each the oracle's camera paths ray by ray (OrcDebugPath) traced again on the GPU
(MsneTraceRays) and by the oracle with and without boxes: the first vertex whose hit differs, with both records and the ray."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
np.set_printoptions(precision=9, linewidth=220)
from oracle import orc
import moonshine_amd.api as api
import hull_rays
seed, far = int(sys.argv[1]), float(sys.argv[2])
oc = orc.Context(threads=8); gc = api.Context()
baked = seed % 3 == 2
world = [hull_rays.hull_scene(c, seed, seed % 2 == 1, baked=baked) for c in (oc, gc)][0]
level = 2 if seed % 2 == 0 else 1
oc.set_exhaustive_search(level)
rs = np.random.default_rng(seed + 4242)
reach = max(float(np.abs(W).max()) for W in world)
sn = [c.create_sensor(24, 16) for c in (oc, gc)]
rays = hull_rays.far_rays(world, seed, far)
if "--like-the-test" in sys.argv:      # the test traces rays before it renders: the same calls in the same order
    import test_gpu_parity as T
    T._check_rays(oc, gc, hull_rays.hull_rays(world, seed, far=1.0)[::7]); T._check_rays(oc, gc, rays)
W = world[int(rs.integers(len(world)))]; ctr = 0.5 * (W.min(0) + W.max(0)); r = max(np.linalg.norm(W - ctr, axis=1).max(), 1e-20)
eye = rs.normal(size=3); eye = eye / np.linalg.norm(eye) * reach * far * 1.7
fwd = ctr - eye; dist = np.linalg.norm(fwd); fwd = fwd / dist
up = np.array([0, 0, 1.0]) if abs(fwd[2]) < 0.9 else np.array([0, 1.0, 0])
films, lenses = [], []
ap = float(rs.choice([0.0, 0.5 * r]))
for c, s_ in zip((oc, gc), sn):
    lens = c.create_lens(c.make_lens(tuple(eye), tuple(fwd), tuple(up), float(2.0 * np.arctan(1.5 * r / dist)), ap, float(dist))); lenses.append(lens)
    c.set_pipeline(samples_per_run=2, max_bounces=3, env_samples_per_bounce=1, mesh_samples_per_bounce=0)
    c.render(s_, lens, launches=2); films.append(c.sensor_data(s_).copy())
print("seed", seed, "far", far, "baked", baked, "level", level, "reach %.4g instance radius %.4g dist %.4g aperture %.4g" % (reach, r, dist, ap), "counters", oc.counters(), gc.counters())
same = ((films[0].view(np.uint32) == films[1].view(np.uint32)) | (np.isnan(films[0]) & np.isnan(films[1]))).all(-1)
ys, xs = np.nonzero(~same)
print("pixels that differ:", list(zip(xs.tolist(), ys.tolist())))
for x, y in list(zip(xs.tolist(), ys.tolist()))[:4]:
    print(" pixel", (x, y), "oracle", films[0][y, x], "hip", films[1][y, x])
    for k in range(4):
        rgb, rec, cnt = oc.debug_path(sn[0], lenses[0], k, x, y)
        if not len(rec):
            print("  k", k, "no hit; rgb", rgb); continue
        R = np.concatenate([rec[:, 8:11], rec[:, 5:8], np.full((len(rec), 1), np.inf, np.float32)], axis=1).astype(np.float32)
        ids, tuv = gc.trace_rays(R)
        for i in range(len(rec)):
            o_ids = (int(rec[i, 0]), int(rec[i, 11]), int(rec[i, 1]))
            if not ids[i, 0] or tuple(int(v) for v in ids[i, 1:4]) != o_ids or not np.array_equal(tuv[i].view(np.uint32), rec[i, 2:5].view(np.uint32)):
                oc.set_exhaustive_search(0); b0 = oc.trace_closest(R[i, :3], R[i, 3:6], 1e30); oc.set_exhaustive_search(level)
                print("  k", k, "vertex", i, "oracle(all)", o_ids, rec[i, 2:5], "| hip", ids[i], tuv[i], "| oracle(bvh)", b0[1], b0[2], "\n     ray", R[i]); break
        else:
            print("  k", k, "same", len(rec), "vertices: a shadow ray or the shading differs; rgb", rgb)
