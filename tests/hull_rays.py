"""Scenes and rays that are hard on a bounding volume of an INSTANCE (the TLAS boxes of both sides, the leaf's sphere of trace.hip's space body): the hit is decided in
instance space, the culling in world space, and the two spaces agree only up to the rounding of the inverse transform and of the transformed origin.  Used on the GPU
(tests/test_gpu_parity.py) and, against the oracle's search without boxes, on the CPU (tests/test_oracle.py)."""
import numpy as np

from moonshine_amd import scenes


def hull_scene(ctx, seed, harsh=False, parts=None, baked=False):
    """14 instances of 4 meshes (an icosphere, a flat quad, two triangle soups; axes scaled 1e-2 ... 10): rotated, mirrored, scaled 1e-3 ... 1e3 per axis (every third),
    sheared (every fourth), carried 0 ... 3e4 away from the origin.  `harsh`: instance 9 is sheared AND scaled 1e-3 ... 1e3 — a transform whose inverse loses six digits,
    where the world-space image of what a ray meets in instance space is anywhere: no culling is possible there, only not culling.  Returns every instance's world-space vertices
    (`parts`: a list that receives every instance's object-space vertices, for hull_move).  `baked`: the transforms are applied to the vertices instead (in double, rounded
    once) and every instance is an identity instance — ONE world BLAS over features of 1e-3 ... 1e3 at coordinates up to 3e4, traced by the kernels without a TLAS level"""
    rs = np.random.default_rng(seed)
    normal = ctx.solid_texture(0.5, 0.5); black = ctx.solid_texture(0.0, 0.0, 0.0)
    grey = ctx.create_material(scenes.LAMBERT, normal, black, color=ctx.solid_texture(0.7, 0.7, 0.7))
    meshes = []; meshes_I = {}
    for k in range(4):
        if k == 0:
            P, I = scenes.icosphere(1)
        elif k == 1:
            P, I = scenes.quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0))
        else:
            P = rs.normal(size=(3 * 24, 3)).astype(np.float32); I = np.arange(3 * 24, dtype=np.uint32).reshape(-1, 3)
        P = (P * np.float32(10.0) ** rs.uniform(-2, 1, (1, 3))).astype(np.float32)
        meshes.append((ctx.create_mesh(P, I), P)); meshes_I[meshes[-1][0]] = I
    world = []
    for k in range(14):
        h, P = meshes[int(rs.integers(len(meshes)))]
        wide = k % 3 == 0 and (harsh or k % 4 != 1)
        M = scenes._rot(rs.normal(size=3), rs.uniform(0, 6.28))[:3, :3] @ np.diag(10.0 ** rs.uniform(-3, 3 if wide else 0.5, 3) * rs.choice([-1.0, 1.0], 3))
        if k % 4 == 1:
            M = M @ (np.eye(3) + np.triu(rs.normal(size=(3, 3)), 1))                # shear
        t = rs.normal(size=3) * (0.0 if k == 0 else 10.0 ** rs.uniform(0, 4.5))
        T = np.zeros((3, 4), np.float32); T[:, :3] = M; T[:, 3] = t
        W = P.astype(np.float64) @ T[:, :3].astype(np.float64).T + T[:, 3].astype(np.float64)
        if baked:
            W = W.astype(np.float32)
            ctx.create_instance([(ctx.create_mesh(W, meshes_I[h]), grey, False)])
            W = W.astype(np.float64)
        else:
            ctx.create_instance([(h, grey, False)], transform=T)
        if parts is not None:
            parts.append(P)
        world.append(W)
    ctx.set_background(np.array([0.5, 0.5, 0.5, 1], np.float32), 1, 1)
    return world


def hull_move(ctxs, seed, parts, world):
    """new transforms of the same kinds for five of the instances, in every context of `ctxs` (transform edits: the product re-fits its TLAS in place, Accel.zig:567-601);
    `world` is updated"""
    rs = np.random.default_rng(seed + 991)
    for k in rs.choice(len(parts), 5, replace=False):
        M = scenes._rot(rs.normal(size=3), rs.uniform(0, 6.28))[:3, :3] @ np.diag(10.0 ** rs.uniform(-3, 2, 3) * rs.choice([-1.0, 1.0], 3))
        if rs.random() < 0.4:
            M = M @ (np.eye(3) + np.triu(rs.normal(size=(3, 3)), 1) * 0.5)
        T = np.zeros((3, 4), np.float32); T[:, :3] = M; T[:, 3] = rs.normal(size=3) * 10.0 ** rs.uniform(0, 4.5)
        for c in ctxs:
            c.set_instance_transform(int(k), T)
        world[k] = parts[k].astype(np.float64) @ T[:, :3].astype(np.float64).T + T[:, 3].astype(np.float64)


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


def far_rays(world, seed, far):
    """(n, 7) rays from `far` times the scene's largest coordinate out (half to all of it, any direction), at the six outermost vertices of every instance and 1e-6 ... 0.3
    instance sizes beside them: origins whose coordinates carry an ulp of up to 0.06 scene sizes — what the instance-space twin of the ray is rounded to, and what every
    world-space volume an instance is culled by has to allow for"""
    rs = np.random.default_rng(seed + 4242)
    reach = max(float(np.abs(W).max()) for W in world)
    rays = []
    for W in world:
        c0 = 0.5 * (W.min(0) + W.max(0)); r = np.linalg.norm(W - c0, axis=1)
        for v in W[np.argsort(r)[-6:]]:
            o = rs.normal(size=3); o = o / np.linalg.norm(o) * reach * far * rs.uniform(0.5, 1.0)
            tgt = v + rs.normal(size=3) * max(r.max(), 1e-30) * rs.choice([0.0, 1e-6, 1e-3, 0.3])
            d = tgt - o; n = np.linalg.norm(d)
            rays.append(np.concatenate([o, d / n, [1e30 if rs.random() < 0.7 else n * rs.uniform(0.9, 1.1)]]))
    rays = np.asarray(rays, np.float32)
    rays[:, 3:6] /= np.linalg.norm(rays[:, 3:6].astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    return rays


def lattice_scale(seed):
    """the scene and its rays times a power of two (exact): 1 for most seeds; 2^-40, 2^40; 2^-62 and 2^62, where products of two coordinates sit at the ends of the range"""
    return np.float32(2.0) ** int(np.random.default_rng(seed + 3).choice([0, 0, 0, 0, -40, 40, -62, 62]))


def lattice_scene(ctx, seed, baked=False, scale=None):
    """axis-aligned unit cubes (twelve triangles each, faces in coordinate planes) at half-integer places: some as identity instances (the merged world BLAS), some
    under 90-degree rotations, mirrors and power-of-two scales with integer translations — every coordinate, product and sum exact in f32"""
    rs = np.random.default_rng(seed)
    normal = ctx.solid_texture(0.5, 0.5); black = ctx.solid_texture(0.0, 0.0, 0.0)
    grey = ctx.create_material(scenes.LAMBERT, normal, black, color=ctx.solid_texture(0.7, 0.7, 0.7))
    c = np.array([[x, y, z] for z in (0, 1) for y in (0, 1) for x in (0, 1)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    I = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], np.uint32)
    S = lattice_scale(seed) if scale is None else np.float32(scale)
    cube = ctx.create_mesh(c * S, I)
    flat = ctx.create_mesh(c[:4] * np.float32(2) * S, np.array([[0, 1, 3], [0, 3, 2]], np.uint32))      # a 2 x 2 square in z = 0
    perms = [np.eye(3)[list(p)] for p in ((0, 1, 2), (1, 2, 0), (2, 0, 1), (1, 0, 2), (0, 2, 1), (2, 1, 0))]
    for k in range(int(rs.integers(4, 9))):
        T = np.zeros((3, 4), np.float32)
        if k % 3 == 0:
            T[:, :3] = np.eye(3)                                                                         # identity: world BLAS
        else:
            T[:, :3] = perms[int(rs.integers(6))] @ np.diag(rs.choice([-1.0, 1.0], 3) * 2.0 ** rs.integers(-1, 2, 3))
        T[:, 3] = rs.integers(-2, 3, 3) * (0.5 if k % 2 else 1.0) * S
        if baked:
            V, J = (c[:4] * np.float32(2) * S, np.array([[0, 1, 3], [0, 3, 2]], np.uint32)) if k % 4 == 3 else (c * S, I)
            ctx.create_instance([(ctx.create_mesh((V.astype(np.float64) @ T[:, :3].astype(np.float64).T + T[:, 3].astype(np.float64)).astype(np.float32), J), grey, False)])
        else:
            ctx.create_instance([(flat if k % 4 == 3 else cube, grey, False)], transform=T)
    ctx.set_background(np.array([0.5, 0.5, 0.5, 1], np.float32), 1, 1)


def lattice_rays(seed, n=1500):
    """origins on the half-integer lattice of [-3, 3]^3, directions out of {-1, -1/2, 0, 1/2, 1}^3 (not normalised: t is in units of d), tmax 1e12 or a small integer or
    half-integer: rays IN face planes, along edges, through corners, starting on faces, ending exactly on them"""
    rs = np.random.default_rng(seed + 5)
    o = rs.integers(-6, 7, (n, 3)) * 0.5
    d = rs.integers(-2, 3, (n, 3)) * 0.5
    d[(d == 0).all(1)] = (1.0, 0.0, 0.0)
    tmax = np.where(rs.random(n) < 0.5, 1e12, rs.integers(1, 9, n) * 0.5)
    rays = np.concatenate([o, d, tmax[:, None]], 1).astype(np.float32)
    rays[:, :3] *= lattice_scale(seed)                                      # (t is then in units of the scale too: d is left alone for half of the rays)
    half = rs.random(n) < 0.5
    rays[half, 3:6] *= lattice_scale(seed)
    rays[~half, 6] = np.minimum(rays[~half, 6].astype(np.float64) * float(lattice_scale(seed)), 3e38).astype(np.float32)
    # a few rays that are not rays: tmax 0, negative, infinite; an infinite or NaN origin or direction component; a zero direction
    k = rs.choice(n, 40, replace=False)
    rays[k[:8], 6] = (0.0, -1.0, np.inf, 0.0, -0.0, np.inf, 1e-45, 3e38)
    rays[k[8:16], 0] = (np.inf, -np.inf, np.nan, np.inf, np.nan, 3e38, -3e38, 1e-45)
    rays[k[16:24], 4] = (np.inf, -np.inf, np.nan, np.inf, np.nan, 3e38, 1e-45, -1e-45)
    rays[k[24:28], 3:6] = 0.0
    rays[k[28:34], 3:6] *= np.float32(1e-30)
    rays[k[34:40], 3:6] *= np.float32(1e30)
    return rays


def face_rays(seed, n=600):
    """rays that LEAVE the faces of the lattice scenes' cubes from almost no distance: origins on a lattice plane in one axis (generic or lattice in the others), moved off
    it by 0 ... 300 ulps (from 0: denormals), 2^-16 or 2^-10, directions anywhere in the half space they move into — the triangle test computes their t to the face as a
    cancellation of O(1) terms (+-2e-8 around a true 0 or -1e-42) and takes some of them"""
    rs = np.random.default_rng(seed + 21)
    rays = np.zeros((n, 7), np.float32)
    for k in range(n):
        p = rs.integers(-6, 7, 3) * 0.5 + rs.random(3) * (rs.random(3) < 0.6)
        ax = int(rs.integers(3)); p[ax] = rs.integers(-6, 7) * 0.5
        sgn = float(rs.choice([-1.0, 1.0]))
        o = p.astype(np.float32)
        off = rs.choice([1.0 / 65536, 2.0 ** -10, 0.0])
        if off == 0.0:
            for _ in range(int(rs.integers(0, 300))):
                o[ax] = np.nextafter(o[ax], np.float32(np.inf * sgn))
        else:
            o[ax] += np.float32(off * sgn)
        d = rs.normal(size=3); d /= np.linalg.norm(d)
        if d[ax] * sgn < 0:
            d = -d
        rays[k, :3] = o; rays[k, 3:6] = d; rays[k, 6] = 1e12
    return rays
