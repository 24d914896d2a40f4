"""usage (GPU box): python tools/film_diff.py FAMILY SEED ... — the scenes of tests/test_gpu_parity.py::test_films_of_hull_and_lattice_scenes rendered by the HIP path and the
oracle with max_bounces 0 .. 5: where the films first differ, in which pixels, with which ray counts (profiles/r05_fuzz_sweeps.txt: the open mismatch of round 5)"""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from oracle import orc
import moonshine_amd.api as api
import hull_rays
fam = sys.argv[1]
for seed in map(int, sys.argv[2:]):
    for mb in range(6):
        films, cnt = [], []
        for C_ in (orc.Context(threads=8), api.Context()):
            rs = np.random.default_rng(seed + 9)
            if fam == "hull":
                world = hull_rays.hull_scene(C_, seed, seed % 2 == 1, baked=seed % 3 == 2)
                wi = int(rs.integers(len(world))); W = world[wi]; ctr = 0.5 * (W.min(0) + W.max(0)); r = max(np.linalg.norm(W - ctr, axis=1).max(), 1e-20)
                eye = ctr + rs.normal(size=3) * r * rs.choice([0.3, 1.5, 4.0]); fwd = ctr - eye + rs.normal(size=3) * r * 0.1
            else:
                S = float(hull_rays.lattice_scale(seed)); S = S if 1e-10 < S < 1e10 else 1.0; wi = -1; r = S
                hull_rays.lattice_scene(C_, seed, baked=seed % 3 == 2, scale=S)
                eye = rs.integers(-6, 7, 3) * 0.5 * S; fwd = rs.integers(-2, 3, 3) * 1.0
                if not fwd.any(): fwd = np.array([1.0, 0, 0])
            up = np.array([0, 0, 1.0]) if abs(fwd[2]) < 0.9 * np.linalg.norm(fwd) else np.array([0, 1.0, 0])
            lens = C_.create_lens(C_.make_lens(tuple(eye), tuple(fwd / np.linalg.norm(fwd)), tuple(up), 0.9, 0.0, 1.0)); sn = C_.create_sensor(24, 16)
            C_.set_pipeline(samples_per_run=2, max_bounces=mb, env_samples_per_bounce=1, mesh_samples_per_bounce=0)
            C_.render(sn, lens, launches=2); films.append(C_.sensor_data(sn).copy()); k = C_.counters(); cnt.append((k["closest_rays"], k["shadow_rays"]))
        A, B = films
        same = ((A.view(np.uint32) == B.view(np.uint32)) | (np.isnan(A) & np.isnan(B))).all(-1)
        ys, xs = np.nonzero(~same)
        print(fam, seed, "instance", wi, "radius %.3g eye %s" % (r, np.round(eye, 4).tolist()), "max_bounces", mb, "| pixels that differ", list(zip(ys.tolist(), xs.tolist()))[:4],
              "| oracle / HIP", [(A[y, x, 0].item(), B[y, x, 0].item()) for y, x in list(zip(ys, xs))[:3]], "| closest, shadow rays: oracle", cnt[0], "HIP", cnt[1])
