"""What is ray ORDER worth to the traversal kernels?  Second-bounce rays of S1 (cosine-distributed about the normals of the primary hits, 2 M of them) are traced by
the probe kernel (k_trace_probe = the same wave loop as k_trace_closest / k_trace_shadow) in four orders: as the wavefront produces them (pixel order), sorted by direction
octant inside every block of 256 (what a per-workgroup sort in k_shade could do), sorted globally by (octant, origin Morton code), and shuffled.  Kernel times come from
rocprofv3 --kernel-trace:  rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 tools/coherence_probe.py ; python tools/coherence_probe.py --report DIR"""
import os, sys, csv, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np

ORDERS = ["pixel order", "octant-sorted per 256", "globally sorted (octant, origin)", "shuffled"]
if "--report" in sys.argv:
    d = sys.argv[sys.argv.index("--report") + 1]
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_trace_probe" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[1:]                                   # [0] = the primary rays
    for k, name in enumerate(ORDERS):
        for any_hit in (0, 1):
            r = rows[2 * k + any_hit]
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            print("%-36s %s  %.3f ms" % (name, "any-hit    " if any_hit else "closest-hit", ms))
    sys.exit(0)

import torch  # noqa
from moonshine_amd import api, scenes
c = api.Context()
s, l = scenes.s1(c, extent=(1920, 1080))
W, H = 1920, 1080
# primary rays through pixel centres of a pinhole at S1's camera (what matters here is the hit points, not the exact lens)
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
# a random direction in the hemisphere about an estimated normal (towards the camera side): cosine-ish, incoherent like a diffuse bounce
COPIES = 8                                            # 8 directions per hit point, interleaved: ~10 M rays, the size of a full-frame bounce
P = np.repeat(P, COPIES, axis=0); prim_d = np.repeat(rays[hit, 3:6], COPIES, axis=0); n = len(P)
nrm = -prim_d + rs.normal(size=(n, 3)) * 0.7; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
w = rs.normal(size=(n, 3)); w /= np.linalg.norm(w, axis=1, keepdims=True); w = w + nrm * 1.2; w /= np.linalg.norm(w, axis=1, keepdims=True)
sec = np.zeros((n, 7), np.float32); sec[:, :3] = P + nrm * 1e-3; sec[:, 3:6] = w; sec[:, 6] = 1e30
octant = (sec[:, 3] > 0).astype(np.int64) | ((sec[:, 4] > 0).astype(np.int64) << 1) | ((sec[:, 5] > 0).astype(np.int64) << 2)
def morton(p):
    q = ((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9) * 1023).astype(np.int64)
    def ex(x):
        x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F; x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
        return x
    return ex(q[:, 0]) | (ex(q[:, 1]) << 1) | (ex(q[:, 2]) << 2)
blk = np.arange(n) // 256
orders = [np.arange(n), np.lexsort((np.arange(n), octant, blk)), np.lexsort((morton(sec[:, :3]), octant)), rs.permutation(n)]
print("%d secondary rays" % n)
for name, idx in zip(ORDERS, orders):
    r = np.ascontiguousarray(sec[idx])
    a, _ = c.trace_rays(r, any_hit=False); b, _ = c.trace_rays(r, any_hit=True)
    print(name, "hits", int((a[:, 0] != 0).sum()), "occluded", int((b[:, 0] != 0).sum()))
