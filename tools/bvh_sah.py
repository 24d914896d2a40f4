"""Surface-area cost of the wide BVHs the library builds (read back with MsneDebugReadBvh): expected node visits and triangle tests of a random ray that hits the
root box = sum of box areas over the root's area, internal nodes and leaf boxes separately, per tree (every BLAS and the TLAS of the scene) — to see how the
builder's knobs ($MSNE_PLOC_RADIUS, $MSNE_MORTON_BITS, $MSNE_SAH_TOP) move the trees, next to the measured rate.   python tools/bvh_sah.py [s1|s2|sky]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from moonshine_amd import api, scenes

scene = sys.argv[1] if len(sys.argv) > 1 else "s1"
c = api.Context()
s, l = scenes.s2(c, extent=(64, 36)) if scene == "s2" else scenes.s1(c, extent=(64, 36))
c.render(s, l, launches=1)
nodes, tris, root, items = c.read_bvh()
N = np.frombuffer(nodes.tobytes(), np.uint8).reshape(-1, 80)
origin = N[:, 0:12].copy().view(np.float32)
ex = N[:, 12:15].astype(np.int32); imask = N[:, 15].astype(np.uint32)
child_base = N[:, 16:20].copy().view(np.uint32)[:, 0]; lmask = N[:, 24].astype(np.uint32)
qlo = N[:, 32:56].reshape(-1, 3, 8).astype(np.float32); qhi = N[:, 56:80].reshape(-1, 3, 8).astype(np.float32)
scale = np.ldexp(1.0, ex - 127).astype(np.float32)


def area(lo, hi):
    d = np.maximum(hi - lo, 0.0)
    return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0])


def tree_cost(r):
    """(sum of internal-node box areas incl. the root, sum of leaf box areas) / root area; boxes as the traversal sees them (dequantised planes)"""
    def child_box(n, s_):
        return origin[n] + qlo[n, :, s_] * scale[n], origin[n] + qhi[n, :, s_] * scale[n]
    used = [s_ for s_ in range(8) if ((int(imask[r]) | int(lmask[r])) >> s_) & 1]
    lo = np.min([child_box(r, s_)[0] for s_ in used], 0); hi = np.max([child_box(r, s_)[1] for s_ in used], 0)
    a_root = area(lo, hi)
    a_int, a_leaf, nn, nl = a_root, 0.0, 1, 0
    stack = [r]
    while stack:
        n = stack.pop(); ci = 0
        for s_ in range(8):
            if (int(imask[n]) >> s_) & 1:
                lo_, hi_ = child_box(n, s_); a_int += area(lo_, hi_); nn += 1; stack.append(int(child_base[n]) + ci); ci += 1
            elif (int(lmask[n]) >> s_) & 1:
                lo_, hi_ = child_box(n, s_); a_leaf += area(lo_, hi_); nl += 1
    return a_int / a_root, a_leaf / a_root, nn, nl


roots = set([int(root)])
# BLAS roots: every node that is nobody's child and not the TLAS root
is_child = np.zeros(len(N), bool)
pc = np.array([bin(i).count("1") for i in range(256)])
for n in range(len(N)):
    k = pc[imask[n]]
    if k: is_child[child_base[n]:child_base[n] + k] = True
for n in np.flatnonzero(~is_child): roots.add(int(n))
print("PLOC radius %s, Morton bits %s, top-down stage over %s clusters" % (os.environ.get("MSNE_PLOC_RADIUS", "4"), os.environ.get("MSNE_MORTON_BITS", "auto"), os.environ.get("MSNE_SAH_TOP", "4096")))
for r in sorted(roots):
    ai, al, nn, nl = tree_cost(r)
    if nn + nl < 4: continue
    print("tree rooted at node %8d%s: %8d nodes %9d leaves   expected node visits %.3f   expected leaf tests %.3f   cost 13 / 22 cycles: %.1f" % (
        r, " (traversal root)" if r == int(root) else "", nn, nl, ai, al, 13.0 * ai + 22.0 * al))
