"""usage (GPU box): python tools/tlas_check.py SEED — the TLAS of tests/hull_rays.py's scene as the product holds it (MsneReadBvh), after the build and after each re-fit of
tests/test_gpu_parity.py::test_rays_at_the_hulls_...: every leaf's dequantised box must hold its instance's world-space vertices, every internal child's box the boxes of
that child's children.  Prints every violation (node, slot, axis, by how much)."""
import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
np.set_printoptions(precision=6, linewidth=200)
import moonshine_amd.api as api
import hull_rays


def boxes(node):
    o = node[0:12].view(np.float32); e = node[12:15]; imask = int(node[15]); cb = int(node[16:20].view(np.uint32)[0]); ib = int(node[20:24].view(np.uint32)[0]); lmask = int(node[24])
    qlo = node[32:56].reshape(3, 8).astype(np.float64); qhi = node[56:80].reshape(3, 8).astype(np.float64)
    sc = np.array([float(np.uint32(int(x) << 23).view(np.float32)) for x in e], np.float64)
    lo = o.astype(np.float64)[:, None] + qlo * sc[:, None]; hi = o.astype(np.float64)[:, None] + qhi * sc[:, None]
    return lo, hi, imask, lmask, cb, ib


def check(gc, world, tag):
    nodes, tris, root, items = gc.read_bvh()
    bad = 0; seen = set()

    def visit(n, depth):
        """-> the instances under node n; every child box is held against the world-space vertices of everything under it"""
        nonlocal bad
        lo, hi, imask, lmask, cb, ib = boxes(nodes[n])
        ci = li = 0
        under = []
        for s in range(8):
            if (imask >> s) & 1:
                sub = visit(cb + ci, depth + 1); ci += 1; what = "child node %d" % (cb + ci - 1)
            elif (lmask >> s) & 1:
                inst = int(items[ib + li]); li += 1; seen.add(inst); sub = [inst]; what = "instance %d" % inst
            else:
                continue
            for inst in sub:
                if inst >= len(world): continue
                W = world[inst]
                for k in range(3):
                    if W[:, k].min() < lo[k, s] or W[:, k].max() > hi[k, s]:
                        bad += 1; print(tag, "node", n, "depth", depth, "slot", s, "axis", k, what, "instance", inst, "sticks out: vertices [%g, %g] box [%g, %g]" % (W[:, k].min(), W[:, k].max(), lo[k, s], hi[k, s]))
            under += sub
        return under
    visit(root, 0)
    print(tag, "TLAS root", root, "instances reached", sorted(seen), "violations", bad, "accel", gc.accel_stats(), flush=True)


seed = int(sys.argv[1])
gc = api.Context()
harsh = seed % 2 == 1; baked = seed % 3 == 2
parts = []

class Both:   # hull_scene wants ONE context that records the object-space parts: use the GPU context and keep the vertices
    pass
world = hull_rays.hull_scene(gc, seed, harsh, parts, baked)
gc.create_sensor(8, 8)
gc.trace_rays(hull_rays.hull_rays(world, seed)); check(gc, world, "built")
hull_rays.hull_move((gc,), seed, parts, world)
gc.trace_rays(hull_rays.hull_rays(world, seed + 1)[::2]); check(gc, world, "moved")
hull_rays.hull_move((gc,), seed + 5, parts, world)
gc.trace_rays(hull_rays.hull_rays(world, seed + 2, far=1.0)[::3]); check(gc, world, "moved2")
