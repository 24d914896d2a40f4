"""What does a SMALL trace launch cost?  Diffuse second-bounce rays of S1 in subsets of 64 ... 2 M rays through the probe kernel (k_trace_probe = the wave loop of
k_trace_closest / k_trace_shadow on the full persistent grid): the duration against the ray count separates the launch floor and the longest ray's latency from throughput.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 tools/tail_probe.py ; python tools/tail_probe.py --report DIR"""
import os, sys, csv, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np

SIZES = [64, 1024, 8192, 45000, 170000, 400000, 800000, 1600000]
if "--report" in sys.argv:
    d = sys.argv[sys.argv.index("--report") + 1]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_trace_probe" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[1:]                                   # [0] = the primary rays
    k = 0
    for n in SIZES:
        for any_hit in (0, 1):
            t = []
            for rep in range(3):
                r = rows[k]; k += 1
                t.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            print("%8d rays %s  %8.1f us (%.1f .. %.1f)   %.2f Grays/s" % (n, "any-hit    " if any_hit else "closest-hit", sorted(t)[1], min(t), max(t), n / sorted(t)[1] / 1e3))
    sys.exit(0)

import torch  # noqa
from moonshine_amd import api, scenes
c = api.Context()
s, l = scenes.s1(c, extent=(1920, 1080))
W, H = 1920, 1080
ys, xs = np.mgrid[0:H, 0:W]
o = np.array([-14.0, 0.0, 6.0]); fwd = np.array([1.0, 0.0, -0.35]); fwd /= np.linalg.norm(fwd); up = np.array([0, 0, 1.0])
u = np.cross(up, -fwd); u /= np.linalg.norm(u); v = np.cross(-fwd, u)
hh = np.tan(0.8 / 2); ww = hh * W / H
d = fwd[None] + ((xs.ravel() + 0.5) / W * 2 - 1)[:, None] * ww * u[None] + (1 - (ys.ravel() + 0.5) / H * 2)[:, None] * hh * v[None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((W * H, 7), np.float32); rays[:, :3] = o; rays[:, 3:6] = d; rays[:, 6] = 1e30
ids, tuv = c.trace_rays(rays, any_hit=False)
hit = ids[:, 0] != 0
P = rays[hit, :3] + rays[hit, 3:6] * tuv[hit, 0:1]
n = len(P)
rs = np.random.default_rng(1)
prim_d = rays[hit, 3:6]
nrm = -prim_d + rs.normal(size=(n, 3)) * 0.7; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
w = rs.normal(size=(n, 3)); w /= np.linalg.norm(w, axis=1, keepdims=True); w = w + nrm * 1.2; w /= np.linalg.norm(w, axis=1, keepdims=True)
sec = np.zeros((n, 7), np.float32); sec[:, :3] = P + nrm * 1e-3; sec[:, 3:6] = w; sec[:, 6] = 1e30
print("%d secondary rays" % n)
for m in SIZES:
    idx = np.sort(rs.choice(n, size=min(m, n), replace=False))     # a thinned bounce: survivors in pixel order
    r = np.ascontiguousarray(sec[idx])
    for any_hit in (False, True):
        for rep in range(3):
            c.trace_rays(r, any_hit=any_hit)
