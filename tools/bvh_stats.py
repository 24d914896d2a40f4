"""Shape of the BVH the library builds for S1: node count, children per node, leaf / internal slots, depth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from moonshine_amd import api, scenes

c = api.Context()
s, l = scenes.s1(c, extent=(64, 36))
c.render(s, l, launches=1)
nodes, tris, root, items = c.read_bvh()
imask = nodes[:, 15].astype(np.uint32); lmask = nodes[:, 24].astype(np.uint32)
pc = np.array([bin(i).count("1") for i in range(256)])
ni, nl = pc[imask], pc[lmask]
print("nodes %d  triangles %d  root %d" % (len(nodes), len(tris), root))
print("children per node: mean %.2f (internal %.2f, leaves %.2f)  histogram %s" % ((ni + nl).mean(), ni.mean(), nl.mean(), np.bincount(ni + nl, minlength=9).tolist()))
print("nodes with only leaves %d, only internal %d, mixed %d" % (int(((ni == 0) & (nl > 0)).sum()), int(((nl == 0) & (ni > 0)).sum()), int(((ni > 0) & (nl > 0)).sum())))
child_base = nodes[:, 16:20].copy().view(np.uint32)[:, 0]
depth = np.zeros(len(nodes), np.int32); order = [root]; depth[root] = 1
for n in order:
    for k in range(ni[n]):
        ch = child_base[n] + k; depth[ch] = depth[n] + 1; order.append(ch)
print("depth: max %d, mean over nodes %.1f" % (depth.max(), depth[depth > 0].mean()))
