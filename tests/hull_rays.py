"""Scenes and rays that are hard on a bounding volume of an INSTANCE (the TLAS boxes of both sides, the leaf's sphere of trace.hip's space body): the hit is decided in
instance space, the culling in world space, and the two spaces agree only up to the rounding of the inverse transform and of the transformed origin.  Used on the GPU
(tests/test_gpu_parity.py) and, against the oracle's search without boxes, on the CPU (tests/test_oracle.py)."""
import numpy as np

from moonshine_amd import scenes


def hull_scene(ctx, seed, harsh=False):
    """14 instances of 4 meshes (an icosphere, a flat quad, two triangle soups; axes scaled 1e-2 ... 10): rotated, mirrored, scaled 1e-3 ... 1e3 per axis (every third),
    sheared (every fourth), carried 0 ... 3e4 away from the origin.  `harsh`: instance 9 is sheared AND scaled 1e-3 ... 1e3 — a transform whose inverse loses six digits,
    where the world-space image of what a ray meets in instance space is anywhere: no culling is possible there, only not culling.  Returns every instance's world-space vertices"""
    rs = np.random.default_rng(seed)
    normal = ctx.solid_texture(0.5, 0.5); black = ctx.solid_texture(0.0, 0.0, 0.0)
    grey = ctx.create_material(scenes.LAMBERT, normal, black, color=ctx.solid_texture(0.7, 0.7, 0.7))
    meshes = []
    for k in range(4):
        if k == 0:
            P, I = scenes.icosphere(1)
        elif k == 1:
            P, I = scenes.quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0))
        else:
            P = rs.normal(size=(3 * 24, 3)).astype(np.float32); I = np.arange(3 * 24, dtype=np.uint32).reshape(-1, 3)
        P = (P * np.float32(10.0) ** rs.uniform(-2, 1, (1, 3))).astype(np.float32)
        meshes.append((ctx.create_mesh(P, I), P))
    world = []
    for k in range(14):
        h, P = meshes[int(rs.integers(len(meshes)))]
        wide = k % 3 == 0 and (harsh or k % 4 != 1)
        M = scenes._rot(rs.normal(size=3), rs.uniform(0, 6.28))[:3, :3] @ np.diag(10.0 ** rs.uniform(-3, 3 if wide else 0.5, 3) * rs.choice([-1.0, 1.0], 3))
        if k % 4 == 1:
            M = M @ (np.eye(3) + np.triu(rs.normal(size=(3, 3)), 1))                # shear
        t = rs.normal(size=3) * (0.0 if k == 0 else 10.0 ** rs.uniform(0, 4.5))
        T = np.zeros((3, 4), np.float32); T[:, :3] = M; T[:, 3] = t
        ctx.create_instance([(h, grey, False)], transform=T)
        world.append(P.astype(np.float64) @ T[:, :3].astype(np.float64).T + T[:, 3].astype(np.float64))
    ctx.set_background(np.array([0.5, 0.5, 0.5, 1], np.float32), 1, 1)
    return world


def hull_rays(world, seed, far=None):
    """(n, 7) rays o, d, tmax.  Per instance, at the six vertices farthest from the centre of its box (the ones that set a bounding sphere): straight at the vertex from
    outside (1e-3 ... 1e5 instance sizes away), along the tangent plane through it, from inside the hull, and from a point ~1e5 from the world origin — or, `far` given, at
    most `far` times the scene's largest coordinate from it (the range the product's baked slack covers: DESIGN.md section 2)"""
    rs = np.random.default_rng(seed + 77)
    reach = max(float(np.abs(W).max()) for W in world)
    rays = []
    for W in world:
        c = 0.5 * (W.min(0) + W.max(0)); r = np.linalg.norm(W - c, axis=1)
        size = max(r.max(), 1e-30)
        for v in W[np.argsort(r)[-6:]]:
            out = (v - c) / max(np.linalg.norm(v - c), 1e-30)
            tang = np.cross(out, rs.normal(size=3)); tang /= max(np.linalg.norm(tang), 1e-30)
            for o in (v + out * size * 10.0 ** rs.uniform(-3, 5),
                      v + tang * size * 10.0 ** rs.uniform(-2, 3),
                      v + (tang + out * 1e-4) * size * 3.0,
                      c + rs.normal(size=3) * size * 0.2,
                      rs.normal(size=3) * 1e5):
                if far is not None and np.abs(o).max() > far * reach:
                    o = o * (far * reach / np.abs(o).max())
                tgt = v + rs.normal(size=3) * size * rs.choice([0.0, 0.0, 1e-6, 1e-3])
                d = tgt - o; n = np.linalg.norm(d)
                if n > 0:
                    rays.append(np.concatenate([o, d / n, [1e12 if rs.random() < 0.7 else n * rs.uniform(0.5, 1.5)]]))
    rays = np.asarray(rays, np.float32)
    rays[:, 3:6] /= np.linalg.norm(rays[:, 3:6].astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    return rays
