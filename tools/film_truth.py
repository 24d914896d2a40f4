"""usage (GPU box): python tools/film_truth.py FAMILY SEED ... — the scenes of tests/test_gpu_parity.py::test_films_of_hull_and_lattice_scenes rendered three ways: the HIP
path, the oracle with its BVH (level 0) and the oracle's search over every triangle (level 2, the contract: orc_bvh.c).  Says who agrees with whom — a seed on which the
oracle's BVH and its own search differ is the ORACLE's miss (round 6: 6204351, then 6226272 and 6240180 from an 88 000-seed CPU sweep), whatever the product does."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from oracle import orc
import moonshine_amd.api as api
import hull_rays
fam = sys.argv[1]
for seed in map(int, sys.argv[2:]):
    films = {}
    for name, C_, level in (("hip", api.Context(), None), ("orc_bvh", orc.Context(threads=8), 0), ("orc_all", orc.Context(threads=8), 2)):
        rs = np.random.default_rng(seed + 9)
        if fam == "hull":
            world = hull_rays.hull_scene(C_, seed, seed % 2 == 1, baked=seed % 3 == 2)
            W = world[int(rs.integers(len(world)))]; ctr = 0.5 * (W.min(0) + W.max(0)); r = max(np.linalg.norm(W - ctr, axis=1).max(), 1e-20)
            eye = ctr + rs.normal(size=3) * r * rs.choice([0.3, 1.5, 4.0]); fwd = ctr - eye + rs.normal(size=3) * r * 0.1
        else:
            S = float(hull_rays.lattice_scale(seed)); S = S if 1e-10 < S < 1e10 else 1.0
            hull_rays.lattice_scene(C_, seed, baked=seed % 3 == 2, scale=S)
            eye = rs.integers(-6, 7, 3) * 0.5 * S; fwd = rs.integers(-2, 3, 3) * 1.0
            if not fwd.any(): fwd = np.array([1.0, 0, 0])
        up = np.array([0, 0, 1.0]) if abs(fwd[2]) < 0.9 * np.linalg.norm(fwd) else np.array([0, 1.0, 0])
        if level is not None: C_.set_exhaustive_search(level)
        lens = C_.create_lens(C_.make_lens(tuple(eye), tuple(fwd / np.linalg.norm(fwd)), tuple(up), 0.9, 0.0, 1.0)); sn = C_.create_sensor(24, 16)
        C_.set_pipeline(samples_per_run=2, max_bounces=5, env_samples_per_bounce=1, mesh_samples_per_bounce=0)
        C_.render(sn, lens, launches=2); k = C_.counters(); films[name] = (C_.sensor_data(sn).copy(), (k["closest_rays"], k["shadow_rays"]))
    def diff(a, b):
        A, B = films[a][0], films[b][0]
        return int((~((A.view(np.uint32) == B.view(np.uint32)) | (np.isnan(A) & np.isnan(B))).all(-1)).sum())
    print(fam, seed, "| pixels that differ: hip vs orc_bvh %d, hip vs orc_all %d, orc_bvh vs orc_all %d | rays hip %s orc_bvh %s orc_all %s" % (
        diff("hip", "orc_bvh"), diff("hip", "orc_all"), diff("orc_bvh", "orc_all"), films["hip"][1], films["orc_bvh"][1], films["orc_all"][1]), flush=True)
