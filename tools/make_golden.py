"""Generates tests/golden/*.npz|json from the CPU oracle (run in the authoring container).

The reference itself cannot run here (no zig / dxc / Vulkan — SURVEY.md header table), so the golden data are
(1) the integer-exact PCG known answers of SURVEY.md Appendix A.1, (2) the reference's own furnace-test scenes
(engine/tests.zig:257-455) rendered by the oracle, and (3) oracle outputs for the pieces the reference never
tests (sampling warps, alias tables, offsetAlongNormal, camera rays, BSDFs, env preprocessing) so that any later
change of the restated algorithm is caught.  A fixture is data: inputs + expected outputs.
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from moonshine_amd import scenes  # noqa: E402
from moonshine_amd.hostinfo import usable_cores

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
orc.build()

# (1) RNG
seeds = [(0, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 0), (5, 1919, 1079)]
rng = {"pcg": {str(a): orc.pcg(a) for a in (0, 1, 0xFFFFFFFF)}, "streams": []}
for s in seeds:
    st, f = orc.rng_floats(*s, 16)
    rng["streams"].append({"seed": list(s), "state0": st, "floats_hex": [bytes(b).hex() for b in f.astype(">f4").view(np.uint8).reshape(-1, 4)], "h_shr8": [int(round(float(x) * 16777216.0)) for x in f]})
json.dump(rng, open(os.path.join(G, "rng.json"), "w"), indent=1)

# (3a) warps, offsets, math
grid = np.stack(np.meshgrid(np.linspace(0, 1, 9), np.linspace(0, 1, 9), indexing="ij"), -1).reshape(-1, 2).astype(np.float32)
sph = orc.square_to_equal_area_sphere(grid)
inv = orc.square_to_equal_area_sphere_inverse(sph)
ps, ns = [], []
for p in (0.0, 1e-3, -1e-3, 1.0, -1.0, 1e3, -1e3):
    for n in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1 / math.sqrt(3),) * 3):
        ps.append((p, p, p)); ns.append(n)
off = orc.offset_along_normal(ps, ns)
xs = np.linspace(-7, 7, 257).astype(np.float32)
x01 = np.linspace(1e-6, 1, 257).astype(np.float32)
at = np.stack([np.cos(np.linspace(0, 2 * math.pi, 64)), np.sin(np.linspace(0, 2 * math.pi, 64))], -1).astype(np.float32)
np.savez(os.path.join(G, "math.npz"), grid=grid, sphere=sph, sphere_inv=inv, off_p=np.array(ps, np.float32), off_n=np.array(ns, np.float32), off_out=off,
         xs=xs, sin=orc.math_probe("sin", xs), cos=orc.math_probe("cos", xs), x01=x01, log=orc.math_probe("log", x01),
         acos=orc.math_probe("acos", np.linspace(-1, 1, 257).astype(np.float32)), atan2_in=at, atan2=orc.math_probe("atan2", at))

# (3b) alias tables
al = {}
rs = np.random.default_rng(1)
for name, w in (("one", [1]), ("two", [1, 1]), ("ramp", [1, 2, 3, 4]), ("zero_one", [0, 1]), ("rand1000", rs.random(1000).astype(np.float32).tolist())):
    a, s, tot = orc.build_alias_table(w)
    al[name] = {"weights": [float(np.float32(x)) for x in w], "alias": a.tolist(), "select_hex": [bytes(b).hex() for b in s.astype(">f4").view(np.uint8).reshape(-1, 4)], "sum": tot}
json.dump(al, open(os.path.join(G, "alias.json"), "w"))

# (3c) camera rays for the two test lenses (tests.zig:293-300, 402-409), zero jitter at the pixel corners of a 32x32 sensor
cams = []
for lens in (orc.make_lens((-3, 0, 0), (1, 0, 0), (0, 0, 1), math.pi / 4), orc.make_lens((0, 0, 0), (1, 0, 0), (0, 0, 1), math.pi / 3)):
    rows = [orc.generate_ray(lens, 32, 32, u, v) for u in (0.0, 0.5, 1.0) for v in (0.0, 0.5, 1.0)]
    cams.append(np.array(rows, np.float32))
np.savez(os.path.join(G, "camera.npz"), lens0=cams[0], lens1=cams[1])

# (3d) BSDF probes
rs = np.random.default_rng(2)
def rdir():
    v = rs.normal(size=3); v /= np.linalg.norm(v); return v.astype(np.float32)
rows = []
for t in (orc.GLASS, orc.LAMBERT, orc.PERFECT_MIRROR, orc.STANDARD_PBR):
    for k in range(24):
        wi, wo, sq = rdir(), rdir(), rs.random(2).astype(np.float32)
        met, rough = float(np.float32(rs.random())), float(np.float32(rs.random()))
        r = orc.bsdf_probe(t, (0.9, 0.6, 0.2), met, rough, 1.5, wi, wo, sq)
        rows.append(np.concatenate([[t, met, rough], wi, wo, sq, [r["pdf"]], r["eval"], r["dir"], [r["sample_pdf"]]]).astype(np.float32))
np.save(os.path.join(G, "bsdf.npy"), np.array(rows, np.float32))

# (2) furnace scenes at reduced sample counts (full counts run in the tests themselves)
for name, builder, spr in (("furnace_white", scenes.furnace_white_sphere, 16), ("furnace_inside", scenes.furnace_inside_sphere, 16)):
    c = orc.Context(threads=usable_cores())
    s, l = builder(c)
    c.set_pipeline(samples_per_run=spr, max_bounces=1024, env_samples_per_bounce=0, mesh_samples_per_bounce=0)
    c.render(s, l)
    np.save(os.path.join(G, name + "_16spp.npy"), c.sensor_data(s))
# S1-mini and Cornell films (all four BSDFs, env + mesh NEE, MIS)
c = orc.Context(threads=usable_cores())
s, l = scenes.s1(c, extent=(64, 36), grid=2, order=2)
c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.render(s, l, launches=4)
np.save(os.path.join(G, "s1_mini_64x36_4spp.npy"), c.sensor_data(s))
c = orc.Context(threads=usable_cores())
s, l = scenes.cornell(c, extent=(48, 48))
c.set_pipeline(samples_per_run=2, max_bounces=8, env_samples_per_bounce=0, mesh_samples_per_bounce=1)
c.render(s, l, launches=2)
np.save(os.path.join(G, "cornell_48_4spp.npy"), c.sensor_data(s))
# env preprocessing
img = scenes.sky_sun_equirect(32, 16)
c = orc.Context()
c.set_background(img, 32, 16)
rgb, lum = c.env()
np.savez(os.path.join(G, "env_32x16.npz"), src=img, rgb=rgb, **{"lum%d" % i: l for i, l in enumerate(lum)})
print("golden fixtures written to", G)

# (4) the import-rule gallery (tests/io_common.py:write_gallery, every rule of World.zig:44-363 in one GLB) rendered by the oracle
import tempfile  # noqa: E402
from tests import io_common as io  # noqa: E402
d = tempfile.mkdtemp()
io.write_gallery(os.path.join(d, "gallery.glb"), os.path.join(d, "sky.exr"))
c = orc.Context(threads=usable_cores())
lens, _ = io.oracle_load(orc, c, os.path.join(d, "gallery.glb"), os.path.join(d, "sky.exr"))
s = c.create_sensor(96, 64)
c.set_pipeline(samples_per_run=4, max_bounces=6, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
c.render(s, lens)
np.save(os.path.join(G, "gallery_96x64_4spp.npy"), c.sensor_data(s))
print("gallery golden written")
