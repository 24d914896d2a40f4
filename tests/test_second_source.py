"""Second-source parity of the shading math: the oracle (oracle/orc_*.h) AND the HIP path (csrc/shade.h, msne_math.h) against
tests/second_source.py — an independent float64 numpy restatement written from the reference's HLSL — on the same inputs,
through the batch probes OrcProbeBatch / MsneShadeProbe.

The oracle and the product share most of their shading text, so "HIP == oracle bit for bit" (tests/test_gpu_parity.py) cannot
see a misreading of material.hlsl / light.hlsl that both carry.  These tests can: (1) value comparison on >= 10^4 random
inputs per function, including grazing, back-facing and total-internal-reflection cases; (2) chi-square tests that the
directions sample() produces are distributed as pdf() says; (3) quadrature of the pdfs; (4) furnace tests in the
reference's own shape (engine/tests.zig:257-344) with the sphere's material swapped for mirror / glass / StandardPBR, and
NEE-on vs NEE-off agreement under a non-constant environment.

Tolerance of (1): the implementations compute in f32, the second source in f64.  A record passes when
|got - ref| <= 1e-5 |ref| + 8 |ref32 - ref| + 1e-7, where ref32 is the second source evaluated in float32: that term is the
rounding noise of the FORMULA ITSELF at that input (e.g. GGX's D near its peak cancels catastrophically in f32 for small alpha
— a property of the reference's expression, not of anybody's restatement).  On the well-conditioned records (ref32 within
1e-6 of ref) the plain 1e-5 relative bound holds and is asserted separately.  Records where a DISCRETE decision (a coin flip,
sq.x < fresnel, a hemisphere test) sits within f32 rounding of its threshold are excluded and counted; at most 0.2 % may be.
"""
import math
import os

import numpy as np
import pytest

from moonshine_amd import scenes
from moonshine_amd.hostinfo import usable_cores

from tests import second_source as ss

N = 20000


# ----------------------------------------------------------------------------------------------------------------------
# inputs
def unit_vectors(rs, n, grazing=0.15):
    """uniform directions on the sphere; a `grazing` fraction has |z| in [1e-4, 2e-2]"""
    v = rs.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    k = int(n * grazing)
    z = 10.0 ** rs.uniform(-4, math.log10(2e-2), k) * rs.choice([-1.0, 1.0], k)
    phi = rs.uniform(0, 2 * math.pi, k); r = np.sqrt(1 - z * z)
    v[:k] = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1)
    return v.astype(np.float32)


def inputs(name, rs, n=N):
    f = np.float32
    if name == "bsdf":
        out = []
        for t in (ss.STANDARD_PBR, ss.LAMBERT, ss.PERFECT_MIRROR, ss.GLASS):
            m = n // 4
            x = np.zeros((m, 15), f)
            x[:, 0] = t
            x[:, 1:4] = rs.uniform(0.02, 1.0, (m, 3))
            x[:, 4] = rs.uniform(0, 1, m); x[: m // 10, 4] = rs.choice([0.0, 1.0], m // 10)          # metalness, with exact 0 / 1
            x[:, 5] = rs.uniform(0.0, 1.0, m); x[: m // 20, 5] = 0.0                                   # roughness, some 0 -> alpha floor 0.001
            x[:, 6] = rs.uniform(1.05, 2.5, m) if t != ss.GLASS else rs.uniform(0.8, 2.4, m)          # ior (< 1: TIR from outside)
            x[:, 7:10] = unit_vectors(rs, m); x[:, 10:13] = unit_vectors(rs, m)
            x[:, 13:15] = rs.uniform(0, 1, (m, 2))
            out.append(x)
        return np.concatenate(out)
    if name in ("env_sample", "equal_area", "triangle", "gaussian", "cosine_hemisphere"):
        x = rs.uniform(0, 1, (n, 2)).astype(f)
        x[:50] = rs.choice([0.0, 0.5, 0.25, 0.75], (50, 2))
        if name == "gaussian":
            x[:, 0] = np.minimum(x[:, 0], np.float32(1 - 2 ** -24))
        return x
    if name in ("env_eval", "env_incoming", "equal_area_inverse", "coordinate_system"):
        x = unit_vectors(rs, n)
        x[:6] = np.float32([[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, -1, 0]])
        return x
    if name == "fresnel_dielectric":
        x = np.zeros((n, 3), f)
        x[:, 0] = rs.uniform(-1.2, 1.2, n); x[:200, 0] = 10.0 ** rs.uniform(-6, -2, 200) * rs.choice([-1, 1], 200)
        x[:, 1] = rs.uniform(0.8, 2.5, n); x[:, 2] = rs.uniform(0.8, 2.5, n)
        return x
    if name == "offset_along_normal":
        x = np.zeros((n, 6), f)
        x[:, :3] = (rs.normal(size=(n, 3)) * 10.0 ** rs.uniform(-4, 4, (n, 1))).astype(f)
        x[: n // 4, :3] = rs.uniform(-1 / 16, 1 / 16, (n // 4, 3))       # both sides of the |p| < 1/32 switch
        x[:8, :3] = 0.0
        x[:, 3:] = unit_vectors(rs, n, 0.0)
        return x
    if name == "area_to_solid_angle":
        x = np.zeros((n, 12), f)
        x[:, 0:3] = rs.normal(size=(n, 3)) * 5; x[:, 3:6] = rs.normal(size=(n, 3)) * 5
        x[:, 6:9] = unit_vectors(rs, n, 0.0); x[:, 9:12] = unit_vectors(rs, n, 0.0)
        return x
    if name == "ggx":
        x = np.zeros((n, 7), f)
        x[:, 0] = np.maximum(rs.uniform(0, 1, n) ** 2, 0.001)
        x[:, 1:4] = unit_vectors(rs, n); x[:, 4:7] = unit_vectors(rs, n)
        return x
    if name == "refract":
        x = np.zeros((n, 7), f)
        x[:, 0:3] = unit_vectors(rs, n)
        x[:, 3:6] = np.float32([0, 0, 1]) * np.sign(x[:, 2:3] + 1e-30)     # as glass uses it: n = faceForward((0,0,1), wi)
        x[: n // 4, 3:6] = unit_vectors(rs, n // 4, 0.0)                    # and arbitrary unit normals
        x[:, 6] = rs.uniform(0.4, 2.5, n)
        return x
    if name == "power_heuristic":
        x = np.zeros((n, 4), f)
        x[:, 0] = rs.integers(1, 65, n); x[:, 2] = rs.integers(1, 65, n)
        x[:, 1] = 10.0 ** rs.uniform(-4, 4, n); x[:, 3] = 10.0 ** rs.uniform(-4, 4, n)
        return x
    if name == "texture":
        x = np.zeros((n, 3), f)
        x[:, 0] = rs.integers(0, 3, n)
        x[:, 1:] = rs.uniform(-3.0, 4.0, (n, 2))                         # repeat addressing: well outside [0, 1] too
        x[:64, 1:] = rs.choice([0.0, 1.0, 0.5, -1.0, 2.0], (64, 2))
        return x
    if name == "frame":
        x = np.zeros((n, 9), f)
        x[:, 0:3] = unit_vectors(rs, n, 0.0)
        s = unit_vectors(rs, n, 0.0); x[:, 3:6] = s
        x[:, 6:9] = unit_vectors(rs, n)
        ok = np.abs((x[:, 0:3] * s).sum(1)) < 0.98                        # a tangent (nearly) parallel to the normal is ill-posed
        return x[ok]
    if name == "mesh_attributes":
        x = np.zeros((n, 51), f)
        c = rs.normal(size=(n, 1, 3)) * 3.0
        x[:, 0:9] = (c + rs.normal(size=(n, 3, 3)) * 10.0 ** rs.uniform(-2, 0.5, (n, 1, 1))).reshape(n, 9)      # triangles of size 0.01 .. 3 around a centre
        x[:, 9:15] = rs.uniform(-2, 3, (n, 6))
        x[: n // 40, 11:15] = np.tile(x[: n // 40, 9:11], (1, 2))                                                 # all three texcoords equal: det == 0 exactly
        p0, p1, p2 = x[:, 0:3].astype(np.float64), x[:, 3:6].astype(np.float64), x[:, 6:9].astype(np.float64)
        gn = np.cross(p0 - p2, p1 - p2); gn /= np.linalg.norm(gn, axis=1, keepdims=True)
        for k in range(3):                                                                                        # shading normals within ~35 degrees of the face normal, not unit length
            v = gn + rs.normal(size=(n, 3)) * 0.3
            x[:, 15 + 3 * k: 18 + 3 * k] = v * rs.uniform(0.5, 1.5, (n, 1))
        b = rs.uniform(0, 1, (n, 2)); flip = b.sum(1) > 1; b[flip] = 1 - b[flip]
        x[:, 24:26] = b; x[:16, 24:26] = [[0, 0], [1, 0], [0, 1], [0.5, 0.5]] * 4
        x[:, 26] = rs.integers(0, 4, n)
        # instance transform: rotation x (non-uniform) scale + translation, and its inverse as the reference computes it on the host in f32
        q = rs.normal(size=(n, 3, 3)); q, _ = np.linalg.qr(q)
        sc = 10.0 ** rs.uniform(-0.7, 0.7, (n, 1, 3)); sc[: n // 2] = sc[: n // 2, :, :1]                         # half uniform scales
        A = q * sc; tr = rs.normal(size=(n, 3)) * 4
        A[:64] = np.eye(3); tr[:64] = 0
        tw = np.concatenate([A, tr[:, :, None]], 2).astype(f)
        Ai = np.linalg.inv(tw[:, :, :3].astype(np.float64)); tm = np.concatenate([Ai, -(Ai @ tw[:, :, 3:].astype(np.float64))], 2).astype(f)
        x[:, 27:39] = tw.reshape(n, 12); x[:, 39:51] = tm.reshape(n, 12)
        return x
    if name == "texture_frame":
        x = np.zeros((n, 13), f)
        x[:, 0:3] = rs.uniform(0, 1, (n, 3)); x[: n // 10, 0:2] = rs.choice([0.0, 1.0], (n // 10, 2)); x[:32, 0:2] = 0.5       # incl. rg whose decoded length exceeds 1 (the saturate), and the flat normal
        nn = unit_vectors(rs, n, 0.0); s0 = unit_vectors(rs, n, 0.0)
        ok = np.abs((nn * s0).sum(1)) < 0.9
        s1 = s0 - nn * (nn * s0).sum(1, keepdims=True); s1 /= np.linalg.norm(s1, axis=1, keepdims=True)
        x[:, 3:6] = nn; x[:, 6:9] = s1; x[:, 9:12] = np.cross(nn, s1)
        x[:, 12] = rs.integers(0, 2, n)
        x[x[:, 12] == 0, 0:3] = (unit_vectors(rs, n, 0.0) * [1, 1, 0.5] + [0, 0, 0.6])[x[:, 12] == 0]                 # three-component textures hold vectors, mostly +z
        return x[ok]
    if name == "camera":
        x = np.zeros((n, 18), f)
        x[:, 0:3] = rs.normal(size=(n, 3)) * 5
        fw = unit_vectors(rs, n, 0.0); up = unit_vectors(rs, n, 0.0)
        ok = np.abs((fw * up).sum(1)) < 0.95
        x[:, 3:6] = fw; x[:, 6:9] = up
        x[:, 9] = rs.uniform(0.2, 2.4, n)                       # vfov (radians)
        x[:, 10] = rs.uniform(0, 0.5, n); x[: n // 4, 10] = 0   # aperture, a quarter pinhole
        x[:, 11] = 10.0 ** rs.uniform(-1, 2, n)                 # focus distance
        x[:, 12] = rs.integers(1, 4097, n); x[:, 13] = rs.integers(1, 4097, n)
        x[:, 14:16] = rs.uniform(0, 1, (n, 2)); x[:, 16:18] = rs.uniform(0, 1, (n, 2)); x[:32, 16:18] = 0.5
        return x[ok]
    raise KeyError(name)


# ----------------------------------------------------------------------------------------------------------------------
# the second source, record for record
def second(name, x, env=None, dtype=np.float64, images=None):
    with ss.precision(dtype), np.errstate(all="ignore"):
        x = np.asarray(x, dtype)
        if name == "texture":
            out = np.zeros((len(x), 4), dtype)
            for t, img in enumerate(images):
                m = x[:, 0] == t
                out[m] = ss.vk_sample_linear(img, x[m, 1:3], mirrored=False)
            return out
        if name == "env_incoming":
            return ss.vk_sample_linear(env.rgb, ss.square_to_equal_area_sphere_inverse(x), mirrored=True)
        if name == "bsdf":
            out = np.zeros((len(x), 8), dtype)
            for t in np.unique(x[:, 0]).astype(int):
                m = x[:, 0] == t; r = x[m]
                d = ss.material(int(t), r[:, 1:4], r[:, 4], r[:, 5], r[:, 6], r[:, 7:10], r[:, 10:13], r[:, 13:15])
                out[m] = np.concatenate([d["pdf"][:, None], d["eval"], d["dir"], d["sample_pdf"][:, None]], 1)
            return out
        if name == "env_sample":
            d, rad, pdf, _ = env.sample(x); return np.concatenate([d, rad, pdf[:, None]], 1)
        if name == "env_eval":
            rad, pdf = env.eval(x); return np.concatenate([rad, pdf[:, None]], 1)
        if name == "equal_area":
            return ss.square_to_equal_area_sphere(x)
        if name == "equal_area_inverse":
            return ss.square_to_equal_area_sphere_inverse(x)
        if name == "triangle":
            return ss.square_to_triangle(x)
        if name == "gaussian":
            return ss.square_to_gaussian(x)
        if name == "cosine_hemisphere":
            return ss.square_to_cosine_hemisphere(x)
        if name == "fresnel_dielectric":
            return ss.fresnel_dielectric(x[:, 0], x[:, 1], x[:, 2])[:, None]
        if name == "offset_along_normal":
            return ss.offset_along_normal(x[:, :3], x[:, 3:]).astype(dtype)
        if name == "coordinate_system":
            a, b = ss.coordinate_system(x); return np.concatenate([a, b], 1)
        if name == "area_to_solid_angle":
            return ss.area_to_solid_angle(x[:, 0:3], x[:, 3:6], x[:, 6:9], x[:, 9:12])[:, None]
        if name == "ggx":
            return np.stack([ss.ggx_D(x[:, 0], x[:, 1:4]), ss.ggx_Lambda(x[:, 0], x[:, 1:4]), ss.ggx_G(x[:, 0], x[:, 1:4], x[:, 4:7])], 1)
        if name == "refract":
            return ss.refract_dir(x[:, 0:3], x[:, 3:6], x[:, 6])
        if name == "power_heuristic":
            return ss.power_heuristic(x[:, 0], x[:, 1], x[:, 2], x[:, 3])[:, None]
        if name == "frame":
            n = x[:, 0:3]; s, t = ss.reorthogonalize(n, x[:, 3:6]); v = x[:, 6:9]
            return np.concatenate([ss.world_to_frame(n, s, t, v), ss.frame_to_world(n, s, t, v)], 1)
        if name == "mesh_attributes":
            fl = x[:, 26].astype(int)
            pos, tc, tf, fr = ss.mesh_attributes(x[:, 0:3], x[:, 3:6], x[:, 6:9], x[:, 9:11], x[:, 11:13], x[:, 13:15], x[:, 15:18], x[:, 18:21], x[:, 21:24],
                                                 x[:, 24:26], (fl & 1) != 0, (fl & 2) != 0, x[:, 27:39].reshape(-1, 3, 4), x[:, 39:51].reshape(-1, 3, 4))
            return np.concatenate([pos, tc, *tf, *fr], 1)
        if name == "texture_frame":
            return np.concatenate(ss.texture_frame(x[:, 0:3], x[:, 3:6], x[:, 6:9], x[:, 9:12], x[:, 12] != 0), 1)
        if name == "camera":
            o, d = ss.camera_generate_ray(x[:, 0:3], x[:, 3:6], x[:, 6:9], x[:, 9], x[:, 10], x[:, 11], x[:, 12], x[:, 13], x[:, 14:16], x[:, 16:18])
            return np.concatenate([o, d], 1)
    raise KeyError(name)


def near_decision(name, x):
    """records where a discrete decision of the formula sits within f32 rounding of its threshold (excluded from the value
    comparison: either outcome is a correct evaluation of the reference's expression in f32)"""
    x = x.astype(np.float64)
    bad = np.zeros(len(x), bool)
    if name == "bsdf":
        t = x[:, 0].astype(int)
        wi, wo = x[:, 7:10], x[:, 10:13]
        pbr = t == ss.STANDARD_PBR
        p_spec = ss.pbr_p_specular(x[:, 4])
        bad |= pbr & (np.abs(x[:, 13] - p_spec) < 1e-6)                      # coinFlipRemap(pSpecularSample, square.x)
        gl = t == ss.GLASS
        with np.errstate(all="ignore"):
            fr = ss.fresnel_dielectric(wo[:, 2], ss.AIR_IOR, x[:, 6])
            up = wo[:, 2] > 0
            eta = np.where(up, ss.AIR_IOR / x[:, 6], x[:, 6] / ss.AIR_IOR)
            sin2t = eta * eta * np.maximum(0, 1 - wo[:, 2] ** 2)
        bad |= gl & ((np.abs(x[:, 13] - fr) < 1e-5) | (np.abs(sin2t - 1) < 1e-5))   # sq.x < fresnel; total internal reflection
    if name == "refract":
        c = (x[:, 0:3] * x[:, 3:6]).sum(1)
        bad |= np.abs(x[:, 6] ** 2 * np.maximum(0, 1 - c * c) - 1) < 1e-5
    if name == "fresnel_dielectric":
        c = np.clip(x[:, 0], -1, 1)
        ei, et = np.where(c > 0, x[:, 1], x[:, 2]), np.where(c > 0, x[:, 2], x[:, 1])
        bad |= np.abs(ei / et * np.sqrt(np.maximum(0, 1 - c * c)) - 1) < 1e-5
    if name == "coordinate_system":
        bad |= np.abs(np.abs(x[:, 0]) - np.abs(x[:, 1])) < 1e-6
    if name == "mesh_attributes":   # getTangentBitangent's det == 0.0 switch: a determinant that is zero in one precision and merely tiny in the other
        ht = (x[:, 26].astype(int) & 1) != 0
        t0 = np.where(ht[:, None], x[:, 9:11], [0.0, 0.0]); t1 = np.where(ht[:, None], x[:, 11:13], [1.0, 0.0]); t2 = np.where(ht[:, None], x[:, 13:15], [1.0, 1.0])
        a, b = t0 - t2, t1 - t2
        det = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
        scale = np.abs(a[:, 0] * b[:, 1]) + np.abs(a[:, 1] * b[:, 0])
        bad |= (det != 0) & (np.abs(det) < 1e-5 * scale)
    if name == "texture_frame":     # a texture normal (nearly) along the tangent leaves Gram-Schmidt nothing to work with: ill-posed, as for "frame"
        two = x[:, 12] != 0
        rg = x[:, 0:2] * 2 - 1
        nts = np.where(two[:, None], np.concatenate([rg, np.sqrt(1 - np.clip((rg * rg).sum(1, keepdims=True), 0, 1))], 1), x[:, 0:3])
        bad |= np.abs(nts[:, 0]) > 0.98 * np.linalg.norm(nts, axis=1)
    return bad


# Outputs that are VECTORS are judged against the vector's length, not component by component: z = sqrt(1 - x^2 - y^2) of a unit
# direction, or 1 - sqrt(1 - u) of a barycentric coordinate, loses relative accuracy in f32 where the component is small while the
# vector as a whole is as accurate as f32 gets.  name -> [(first, last+1, fixed scale or None for the vector's own length)]
VECTOR_GROUPS = {"texture": [(0, 4, 1.0)], "env_incoming": [(0, 3, None)], "bsdf": [(4, 7, None)], "env_sample": [(0, 3, None)], "equal_area": [(0, 3, None)], "cosine_hemisphere": [(0, 3, None)],
                 "coordinate_system": [(0, 3, None), (3, 6, None)], "refract": [(0, 3, None)], "frame": [(0, 3, None), (3, 6, None)],
                 "triangle": [(0, 2, 1.0)], "gaussian": [(0, 2, None)], "equal_area_inverse": [(0, 2, 1.0)], "offset_along_normal": [(0, 3, None)],
                 "mesh_attributes": [(0, 3, None), (3, 5, None), (5, 8, 1.0), (8, 11, 1.0), (11, 14, 1.0), (14, 17, 1.0), (17, 20, 1.0), (20, 23, 1.0)],
                 "texture_frame": [(0, 3, 1.0), (3, 6, 1.0), (6, 9, 1.0)], "camera": [(0, 3, None), (3, 6, 1.0)]}


def magnitude(name, ref):
    m = np.abs(ref).copy()
    for a, b, scale in VECTOR_GROUPS.get(name, []):
        m[:, a:b] = np.linalg.norm(ref[:, a:b], axis=1, keepdims=True) if scale is None else scale
    return m


def check_values(name, got, x, env=None, rel=1e-5, images=None):
    ref = second(name, x, env, np.float64, images)
    ref32 = second(name, x, env, np.float32, images).astype(np.float64)
    got = got.astype(np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    skip = near_decision(name, x)
    if name == "env_sample":   # a different texel chosen where a coin flip sits on its threshold
        skip |= (np.abs(got[:, 6] - ref[:, 6]) > 1e-5 * np.abs(ref[:, 6])) & (np.abs(got[:, :3] - ref[:, :3]).max(1) < 3.0 / env.size)
    # ("texture" / "env_incoming": where floor(u - 1/2) sits on an integer the texel pair that is blended is a rounding matter, but the blend is
    #  continuous there: nothing to exclude)
    if name in ("env_eval",):    # the direction falls on a texel border of the equal-area map
        uv = ss.square_to_equal_area_sphere_inverse(x.astype(np.float64)) * env.size
        skip |= (np.abs(uv - np.round(uv)) < 2e-4).any(1)
    assert skip.mean() <= 2e-3, "%s: %.3f %% of the records sit on a decision threshold" % (name, 100 * skip.mean())
    finite = np.isfinite(ref).all(1) & np.isfinite(ref32).all(1)
    assert np.isfinite(got[finite & ~skip]).all(), "%s: non-finite output where the reference formula is finite" % name
    use = finite & ~skip
    noise = np.abs(ref32 - ref)
    if name == "env_sample":
        # The mip descent rescales the random number at every level ((r - p) / (1 - p) or r / p): choosing a texel of probability P leaves
        # the position inside it with ~24 - log2(1/P) bits, whatever the order of the f32 operations.  |ref32 - ref| samples that loss once;
        # the sensitivity to a one-ulp change of the INPUT bounds it: both are rounding noise of the reference's formula at that input.
        x32 = x.astype(np.float32)
        for toward in (np.float32(2.0), np.float32(-1.0)):
            moved = second(name, np.nextafter(x32, toward), env, np.float64, images)
            same_texel = np.abs(moved[:, 6] - ref[:, 6]) <= 1e-9 * np.abs(ref[:, 6])
            noise = np.maximum(noise, np.where(same_texel[:, None], np.abs(moved - ref), 0.0))
    mag = magnitude(name, ref)
    tol = rel * mag + 8.0 * noise + 1e-7
    err = np.abs(got - ref)
    badrec = (err > tol).any(1) & use
    if badrec.any():
        i = int(np.flatnonzero(badrec)[0])
        raise AssertionError("%s: %d of %d records differ from the second source; first: in=%s got=%s ref=%s ref32=%s"
                             % (name, int(badrec.sum()), int(use.sum()), x[i], got[i], ref[i], ref32[i]))
    # on well-conditioned records the plain relative bound holds
    well = use & (noise <= 1e-6 * mag + 1e-12).all(1)
    relerr = (err[well] / (mag[well] + 1e-6)).max() if well.any() else 0.0
    assert well.sum() >= 0.5 * len(x) or name in ("bsdf", "ggx"), "%s: only %d well-conditioned records" % (name, int(well.sum()))
    assert relerr <= rel, "%s: max relative error %.3g on %d well-conditioned records" % (name, relerr, int(well.sum()))
    return dict(records=int(use.sum()), well=int(well.sum()), max_rel_err=float(relerr), skipped=int(skip.sum()))


STATELESS = ["bsdf", "equal_area", "equal_area_inverse", "triangle", "gaussian", "cosine_hemisphere", "fresnel_dielectric",
             "offset_along_normal", "coordinate_system", "area_to_solid_angle", "ggx", "refract", "power_heuristic", "frame",
             "mesh_attributes", "texture_frame", "camera"]
ENV = ["env_sample", "env_eval", "env_incoming", "texture"]


def probe_textures(ctx):
    """three material textures for the sampler probe: 13x7 float RGBA, 8x8 sRGB bytes, a 5x1 strip (one-texel-high: both rows are the same)"""
    rs = np.random.default_rng(99)
    a = rs.random((7, 13, 4)).astype(np.float32)
    b8 = rs.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    c = rs.random((1, 5, 4)).astype(np.float32)
    h = [ctx.create_texture(a, 13, 7, "r32g32b32a32_sfloat"), ctx.create_texture(b8, 8, 8, "r8g8b8a8_srgb"), ctx.create_texture(c, 5, 1, "r32g32b32a32_sfloat")]
    lin = b8.astype(np.float64) / 255.0
    srgb = np.where(lin <= 0.04045, lin / 12.92, ((lin + 0.055) / 1.055) ** 2.4); srgb[..., 3] = lin[..., 3]      # MaterialManager.zig: r8g8b8a8_srgb, alpha linear
    return h, [a.astype(np.float64), srgb, c.astype(np.float64)]


class OrcProbe:
    def __init__(self, orc):
        self.orc = orc
        self.ctx = orc.Context()
        img = scenes.sky_sun_equirect()
        self.ctx.set_background(img, img.shape[1], img.shape[0])
        rgb, lum = self.ctx.env()
        self.env_textures = (rgb, lum)
        self.tex_handles, self.tex_images = probe_textures(self.ctx)

    def __call__(self, name, x):
        return self.orc.probe(name, x, self.ctx)


class GpuProbe:
    def __init__(self, api):
        self.ctx = api.Context()
        img = scenes.sky_sun_equirect()
        self.ctx.set_background(img, img.shape[1], img.shape[0])
        rgb, lum = self.ctx.env()
        self.env_textures = (rgb, lum)
        self.tex_handles, self.tex_images = probe_textures(self.ctx)

    def __call__(self, name, x):
        fn, wi, wo = ss.PROBES[name]
        return self.ctx.shade_probe(fn, wi, wo, x)


def env_of(probe):
    rgb, lum = probe.env_textures
    # the pyramid is the reference's fold.hlsl applied to level 0: rebuilt here in f64 and compared before use
    mine = ss.fold_pyramid(np.asarray(lum[0], np.float64))
    assert len(mine) == len(lum)
    for a, b in zip(mine, lum):
        assert np.allclose(a, b, rtol=2e-6), "luminance pyramid is not the 2x2 sum fold of its level 0"
    assert np.allclose(lum[0], ss.luminance(np.asarray(rgb, np.float64)[..., :3]), rtol=2e-6, atol=1e-9)     # luminance.hlsl
    return ss.EnvMap(rgb, [np.asarray(l, np.float32) for l in lum])    # the f32 textures the shader reads


def run_value_checks(probe, names):
    rs = np.random.default_rng(20240607)
    env = env_of(probe)
    report = {}
    for name in names:
        x = inputs(name, rs)
        if name == "texture":
            x[:, 0] = np.asarray(probe.tex_handles, np.float32)[x[:, 0].astype(int)]          # texture handles of this context
            x2 = x.copy(); x2[:, 0] = np.searchsorted(np.asarray(probe.tex_handles), x[:, 0])
            report[name] = check_values(name, probe(name, x), x2, env, rel=1e-4, images=probe.tex_images)
            continue
        # bilinear lookups: the blend weight frac(u * size - 1/2) is formed in f32 and carries ~2^-23 * u * size of absolute error (3e-5 at size 256),
        # which the blend multiplies by the texel contrast (the sun's texels differ 50-fold): 1e-4 of the vector's magnitude there, 1e-5 everywhere else
        report[name] = check_values(name, probe(name, x), x, env, rel=1e-4 if name == "env_incoming" else 1e-5)
        if name == "offset_along_normal":   # integer arithmetic on bit patterns: exact
            ref = ss.offset_along_normal(x[:, :3], x[:, 3:])
            assert np.array_equal(probe(name, x).view(np.uint32), ref.view(np.uint32))
    return report


# ----------------------------------------------------------------------------------------------------------------------
# (1) values
def test_oracle_matches_second_source_values(orc):
    rep = run_value_checks(OrcProbe(orc), STATELESS + ENV)
    assert all(r["records"] >= 10000 for k, r in rep.items() if k != "frame"), rep
    print(rep)


@pytest.mark.gpu
def test_hip_matches_second_source_values(gpu_api):
    rep = run_value_checks(GpuProbe(gpu_api), STATELESS + ENV)
    assert all(r["records"] >= 10000 for k, r in rep.items() if k != "frame"), rep


def test_second_source_self_consistency():
    """properties the HLSL implies, checked on the second source alone (it must be trustworthy before it judges others)"""
    rs = np.random.default_rng(1)
    sq = rs.uniform(0, 1, (100000, 2))
    d = ss.square_to_equal_area_sphere(sq)
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-12
    assert np.abs(ss.square_to_equal_area_sphere_inverse(d) - sq).max() < 1e-7          # inverse o forward = id (PI is an f32 literal)
    # equal-area: uniform squares -> uniform directions (chi-square over 8 x 16 cells of (cos theta, phi))
    h, _, _ = np.histogram2d(d[:, 2], np.arctan2(d[:, 1], d[:, 0]), bins=(8, 16), range=((-1, 1), (-math.pi, math.pi)))
    e = len(sq) / h.size
    assert ((h - e) ** 2 / e).sum() < h.size + 5 * math.sqrt(2 * h.size)
    # Fresnel: F(normal incidence) = ((n-1)/(n+1))^2, F(grazing) = 1, symmetric under swapping media with the refracted angle
    n = 1.5
    assert abs(ss.fresnel_dielectric(1.0, 1.0, n) - ((n - 1) / (n + 1)) ** 2) < 1e-12
    assert abs(ss.fresnel_dielectric(1e-9, 1.0, n) - 1.0) < 1e-6
    ci = rs.uniform(0.05, 1, 1000); ct = np.sqrt(1 - (1 - ci * ci) / n ** 2)
    assert np.abs(ss.fresnel_dielectric(ci, 1.0, n) - ss.fresnel_dielectric(ct, n, 1.0)).max() < 1e-12
    # glass: Snell's law for the refracted direction, f |cos| / pdf == 1 on both branches
    wo = ss.normalize(rs.normal(size=(1000, 3)))
    sqg = rs.uniform(0, 1, (1000, 2))
    wi, pdf = ss.glass_sample(n, wo, sqg)
    ok = pdf > 0
    assert np.abs(ss.glass_eval(n, wi, wo)[ok, 0] * np.abs(wi[ok, 2]) / pdf[ok] - 1).max() < 1e-12
    tr = ok & (wi[:, 2] * wo[:, 2] < 0)
    eta_o, eta_i = np.where(wo[tr, 2] > 0, ss.AIR_IOR, n), np.where(wo[tr, 2] > 0, n, ss.AIR_IOR)
    assert np.abs(eta_o * np.sqrt(1 - wo[tr, 2] ** 2) - eta_i * np.sqrt(1 - wi[tr, 2] ** 2)).max() < 1e-9
    # StandardPBR: reciprocity of eval; eval >= 0; the sampled direction's sample_pdf == pdf(dir) where positive
    for met, rough in ((0.0, 0.5), (1.0, 0.3), (0.5, 0.8)):
        a, b = ss.normalize(np.abs(rs.normal(size=(2000, 3)))), ss.normalize(np.abs(rs.normal(size=(2000, 3))))
        col = np.array([0.9, 0.6, 0.2])
        al = ss.alpha_from_roughness(rough)
        f_ab, f_ba = ss.pbr_eval(col, met, al, 1.5, a, b), ss.pbr_eval(col, met, al, 1.5, b, a)
        assert np.abs(f_ab - f_ba).max() <= 1e-12 * np.abs(f_ab).max() and (f_ab >= 0).all()
        wi, sp = ss.pbr_sample(al, met, b, rs.uniform(0, 1, (2000, 2)))
        pos = sp > 0
        assert np.abs(ss.pbr_pdf(al, met, wi[pos], b[pos]) - sp[pos]).max() <= 1e-9 * sp[pos].max()


# ----------------------------------------------------------------------------------------------------------------------
# (2) chi-square: the directions sample() returns are distributed as pdf() says
def sphere_bins(d, n_cos, n_phi):
    ic = np.clip(((d[:, 2] + 1) / 2 * n_cos).astype(int), 0, n_cos - 1)
    ip = np.clip(((np.arctan2(d[:, 1], d[:, 0]) + math.pi) / (2 * math.pi) * n_phi).astype(int), 0, n_phi - 1)
    return ic * n_phi + ip


def integrate_over_bins(f, n_cos, n_phi, sub=24):
    """midpoint rule, sub x sub points per (cos theta, phi) bin -> integral of f over each bin (solid angle measure)"""
    nc, np_ = n_cos * sub, n_phi * sub
    c = (np.arange(nc) + 0.5) / nc * 2 - 1
    p = (np.arange(np_) + 0.5) / np_ * 2 * math.pi - math.pi
    C, P = np.meshgrid(c, p, indexing="ij")
    r = np.sqrt(1 - C * C)
    d = np.stack([r * np.cos(P), r * np.sin(P), C], -1).reshape(-1, 3)
    v = f(d).reshape(n_cos, sub, n_phi, sub).sum((1, 3)) * (2.0 / nc) * (2 * math.pi / np_)
    return v.reshape(-1)


def chi_square(obs, exp, min_expected=10.0):
    """Pearson statistic with small cells pooled; returns (statistic, dof)"""
    order = np.argsort(exp)
    o, e = obs[order].astype(np.float64), exp[order].astype(np.float64)
    small = e < min_expected
    o = np.concatenate([[o[small].sum()], o[~small]]); e = np.concatenate([[e[small].sum()], e[~small]])
    if e[0] < min_expected:
        o, e = o[1:], e[1:]
    return float(((o - e) ** 2 / e).sum()), len(e) - 1


PBR_POINTS = [(0.3, 1.0), (0.5, 0.5), (0.8, 0.0)]        # (roughness, metalness)


def run_bsdf_chi_square(probe):
    rs = np.random.default_rng(7)
    n = 400000
    n_cos, n_phi = 16, 24
    out = []
    for rough, met in PBR_POINTS:
        for wo in (np.array([math.sin(0.7), 0.0, math.cos(0.7)]), np.array([0.3, -0.5, -0.81]) / np.linalg.norm([0.3, -0.5, -0.81])):
            x = np.zeros((n, 15), np.float32)
            x[:, 0] = ss.STANDARD_PBR; x[:, 1:4] = (0.9, 0.6, 0.2); x[:, 4] = met; x[:, 5] = rough; x[:, 6] = 1.5
            x[:, 7:10] = (0, 0, 1); x[:, 10:13] = wo.astype(np.float32); x[:, 13:15] = rs.uniform(0, 1, (n, 2))
            got = probe("bsdf", x)
            d, sp = got[:, 4:7].astype(np.float64), got[:, 7]
            kept = sp > 0     # integrator.hlsl:159: a sample with pdf 0 ends the path.  (A NEGATIVE pdf — GGX::sample with w_o.h < 0 — is
            #                   the reference's own wart and is counted with the kept ones: its direction is distributed like the rest)
            kept |= sp < 0
            al = float(ss.alpha_from_roughness(rough))
            wo64 = np.float32(wo).astype(np.float64)
            pdf_bins = integrate_over_bins(lambda w: np.maximum(ss.pbr_pdf(al, met, w, np.broadcast_to(wo64, w.shape)), 0.0), n_cos, n_phi)
            total = pdf_bins.sum()
            assert 0.5 < total <= 1.0 + 2e-3, "integral of StandardPBR.pdf over the sphere = %g" % total           # (3) quadrature
            obs = np.bincount(sphere_bins(d[kept], n_cos, n_phi), minlength=n_cos * n_phi)
            lost = n - int(kept.sum())                                             # specular reflections that leave the hemisphere
            stat, dof = chi_square(np.append(obs, lost), np.append(pdf_bins * n, (1.0 - total) * n))
            out.append((rough, met, stat, dof, total))
            assert stat < dof + 6 * math.sqrt(2 * dof), "StandardPBR roughness %g metalness %g: chi2 %.1f for %d dof" % (rough, met, stat, dof)
    return out


def run_env_chi_square(probe):
    rs = np.random.default_rng(11)
    env = env_of(probe)
    n = 1000000
    got = probe("env_sample", rs.uniform(0, 1, (n, 2)).astype(np.float32))
    d, pdf = got[:, :3].astype(np.float64), got[:, 6].astype(np.float64)
    S = env.size
    uv = ss.square_to_equal_area_sphere_inverse(d)
    ix, iy = np.clip((uv[:, 0] * S).astype(int), 0, S - 1), np.clip((uv[:, 1] * S).astype(int), 0, S - 1)
    # expected texel probabilities = luminance / integral (light.hlsl:66); pooled over 8x8 texel blocks, the sun's block finer
    p = env.lum[0].astype(np.float64) / env.lum[0].astype(np.float64).sum()
    B = 8
    obs = np.bincount((iy // B) * (S // B) + ix // B, minlength=(S // B) ** 2)
    exp = p.reshape(S // B, B, S // B, B).sum((1, 3)).reshape(-1) * n
    stat, dof = chi_square(obs, exp)
    assert stat < dof + 6 * math.sqrt(2 * dof), "EnvMap.sample: chi2 %.1f for %d dof" % (stat, dof)
    hot = np.argsort(p.reshape(-1))[-64:]                                           # the 64 brightest texels one by one
    obs_t = np.bincount(iy * S + ix, minlength=S * S)[hot]
    stat2, dof2 = chi_square(obs_t, p.reshape(-1)[hot] * n)
    assert stat2 < dof2 + 6 * math.sqrt(2 * dof2), "EnvMap.sample (sun texels): chi2 %.1f for %d dof" % (stat2, dof2)
    # the pdf a sample reports is the pdf eval() gives for its direction, and the pdfs of all texels integrate to 1 (quadrature)
    _, pe = env.eval(d[:20000])
    agree = np.abs(pe - pdf[:20000]) <= 1e-5 * pdf[:20000]
    assert agree.mean() > 0.995
    centres = (np.stack(np.meshgrid(np.arange(S), np.arange(S), indexing="xy"), -1).reshape(-1, 2) + 0.5) / S
    got_eval = probe("env_eval", ss.square_to_equal_area_sphere(centres).astype(np.float32))
    assert abs(got_eval[:, 3].astype(np.float64).sum() * 4 * ss.PI / (S * S) - 1.0) < 1e-4
    return stat, dof, stat2, dof2


def test_oracle_sampling_distributions(orc):
    p = OrcProbe(orc)
    run_bsdf_chi_square(p)
    run_env_chi_square(p)


@pytest.mark.gpu
def test_hip_sampling_distributions(gpu_api):
    p = GpuProbe(gpu_api)
    run_bsdf_chi_square(p)
    run_env_chi_square(p)


# ----------------------------------------------------------------------------------------------------------------------
# (4) furnace tests in the reference's shape (engine/tests.zig:257-344) with other materials, on the oracle and on the HIP path
def render(ctx, builder, pipe, launches=1, **kw):
    s, l = builder(ctx, **kw)
    ctx.set_pipeline(**pipe)
    ctx.render(s, l, launches=launches)
    return ctx.sensor_data(s)[..., :3].astype(np.float64)


def run_furnace(make_ctx):
    P = dict(samples_per_run=256, max_bounces=1024, env_samples_per_bounce=0, mesh_samples_per_bounce=0)   # tests.zig:330-335 at 256 spp
    # perfect mirror: f |cos| / pdf is exactly 1 and a convex sphere is hit once -> every pixel is the environment's 1.0
    img = render(make_ctx(), scenes.furnace_sphere, P, kind=scenes.PERFECT_MIRROR)
    assert np.abs(img - 1.0).max() <= 1e-5, "mirror furnace: max |x - 1| = %g" % np.abs(img - 1.0).max()
    # glass: f |cos| / pdf == 1 on both branches, so a path is worth exactly 1 when it leaves the sphere and 0 when the bounce cap
    # ends it inside (integrator.hlsl:126-128).  With max_bounces = 1 a path is lost exactly when it refracts in (1 - F) and is
    # reflected at its first internal hit (F; same angle on a sphere): every pixel is then k / spp for an integer k, and the image
    # mean is 1 - (1/A) * integral of (1 - F) F over the sphere's disc, F = Fresnel::dielectric from the SECOND SOURCE — a full
    # render that measures the Fresnel branch probabilities.  (An uncapped glass furnace does not converge to 1 in any finite
    # sample: Russian roulette's 1/0.95 weights on long total-internal-reflection chains inside the faceted sphere are heavy-tailed.)
    spp = 1024
    G = dict(samples_per_run=spp, max_bounces=1, env_samples_per_bounce=0, mesh_samples_per_bounce=0)
    for ior in (1.5, 2.4):
        img = render(make_ctx(), scenes.furnace_sphere, G, kind=scenes.GLASS, ior=ior, order=6)   # order 6: the facets' tilt adds 10 % / 4 % / 1 % to the loss at order 4 / 5 / 6
        k = img * spp
        assert np.abs(k - np.round(k)).max() < 0.05 and img.min() >= 0.0 and img.max() <= 1.0 + 1e-6, "glass furnace: a path is worth neither 0 nor 1"
        r = (np.arange(200000) + 0.5) / 200000 * math.tan(math.asin(1.0 / 3.0))          # tan(psi) of the primary ray, lens 3 radii away
        sin_t = np.minimum(3.0 * r / np.sqrt(1 + r * r), 1.0)
        F = ss.fresnel_dielectric(np.sqrt(1 - sin_t ** 2), ss.AIR_IOR, ior)
        lost = float(((1 - F) * F * 2 * math.pi * r).sum() * (r[1] - r[0])) / (2 * math.tan(math.pi / 8)) ** 2
        got = 1.0 - img.mean()
        assert abs(got - lost) < 0.03 * lost + 3 * math.sqrt(lost / (img.shape[0] * img.shape[1] * spp)), "glass furnace ior %g: %g of the energy lost, Fresnel predicts %g" % (ior, got, lost)
        # the same with light sampling switched on: delta materials take no light samples, nothing may change
        img2 = render(make_ctx(), scenes.furnace_sphere, dict(G, env_samples_per_bounce=1), kind=scenes.GLASS, ior=ior, order=6)
        assert np.array_equal(img2, img)
    # StandardPBR: the sphere is convex, so a camera path scatters once and then sees the furnace: a pixel's expectation is the
    # directional albedo  rho(w_o) = integral over the upper hemisphere of f(w_i, w_o) |cos w_i|  of the reference's BSDF
    # (which adds the full diffuse lobe to the specular one: white colour gives rho > 1).  rho comes from the SECOND SOURCE's eval by
    # quadrature; the render estimates it with the code under test's sample / pdf / eval: three colour channels x three materials.
    def albedo(color, met, rough, ior, cos_o):
        nc, nphi = 256, 512
        c = (np.arange(nc) + 0.5) / nc; ph = (np.arange(nphi) + 0.5) / nphi * 2 * math.pi
        C, PH = np.meshgrid(c, ph, indexing="ij"); R = np.sqrt(1 - C * C)
        wi = np.stack([R * np.cos(PH), R * np.sin(PH), C], -1).reshape(-1, 3)
        out = []
        for co in cos_o:
            wo = np.broadcast_to(np.array([math.sqrt(1 - co * co), 0.0, co]), wi.shape)
            f = ss.pbr_eval(np.asarray(color, np.float64), met, ss.alpha_from_roughness(rough), ior, wi, wo)
            out.append((f * wi[:, 2:3]).sum(0) * (1.0 / nc) * (2 * math.pi / nphi))
        return np.array(out)
    nr = 96
    r = (np.arange(nr) + 0.5) / nr * math.tan(math.asin(1.0 / 3.0))
    cos_o = np.sqrt(np.maximum(1 - (3.0 * r / np.sqrt(1 + r * r)) ** 2, 0.0))
    area = (2 * math.tan(math.pi / 8)) ** 2
    for rough, met, color in ((1.0, 0.0, (1.0, 1.0, 1.0)), (0.3, 1.0, (0.9, 0.6, 0.2)), (0.6, 0.5, (0.8, 0.7, 0.6))):
        img = render(make_ctx(), scenes.furnace_sphere, dict(P, samples_per_run=512), kind=scenes.STANDARD_PBR, roughness=rough, metalness=met, color=color, order=6)
        rho = albedo(color, met, rough, 1.5, cos_o)                                   # (nr, 3)
        want = 1.0 + ((rho - 1.0) * (2 * math.pi * r * (r[1] - r[0]))[:, None]).sum(0) / area
        got = img.reshape(-1, 3).mean(0)
        assert np.abs(got - want).max() < 0.01, "PBR furnace (roughness %g, metalness %g): image mean %s, albedo quadrature of the second source %s" % (rough, met, got, want)
        assert np.abs(img[0, 0] - 1.0).max() <= 1e-5        # a corner pixel sees only the environment
    # light sampling must not change the expectation (MIS weights sum to one): StandardPBR under the sky+sun environment
    Q = dict(samples_per_run=1, max_bounces=4, mesh_samples_per_bounce=0)
    a = render(make_ctx(), scenes.furnace_sphere, dict(Q, env_samples_per_bounce=0), launches=4096, kind=scenes.STANDARD_PBR, roughness=0.6, metalness=0.3, color=(0.8, 0.7, 0.6), env="sky", extent=(16, 16))
    b = render(make_ctx(), scenes.furnace_sphere, dict(Q, env_samples_per_bounce=1), launches=1024, kind=scenes.STANDARD_PBR, roughness=0.6, metalness=0.3, color=(0.8, 0.7, 0.6), env="sky", extent=(16, 16))
    sa, sb = a[4:12, 4:12].mean(), b[4:12, 4:12].mean()
    assert abs(sa - sb) / sb < 0.05, "NEE on / off under the sky environment: %g vs %g" % (sa, sb)


def test_oracle_material_furnaces(orc):
    run_furnace(lambda: orc.Context(threads=usable_cores()))


@pytest.mark.gpu
def test_hip_material_furnaces(gpu_api):
    run_furnace(lambda: gpu_api.Context())


# ----------------------------------------------------------------------------------------------------------------------
# (5) direct lighting against quadrature: a Lambert floor under (a) an emissive quad sampled as a mesh light, (b) the sky+sun environment.
# With max_bounces = 0 a path is: camera -> floor (light sample with MIS) -> BSDF-sampled ray (emission / environment with the
# complementary MIS weight) -> end (integrator.hlsl:108-135,168-181), i.e. exactly the direct lighting, whose expectation is
# rho / pi * integral of L cos(theta) dw — computed here by plain quadrature over the emitter's area / the environment's texels.
def direct_light_scene(ctx, kind, extent=(24, 24)):
    normal = ctx.solid_texture(0.5, 0.5); black = ctx.solid_texture(0.0, 0.0, 0.0)
    floor = ctx.create_material(scenes.LAMBERT, normal, black, color=ctx.solid_texture(0.6, 0.5, 0.4))
    P, I = scenes.quad((-40, -40, 0), (40, -40, 0), (40, 40, 0), (-40, 40, 0))
    ctx.create_instance([(ctx.create_mesh(P, I), floor, False)])
    if kind == "mesh":
        # a 1 x 2 emitter 1.5 above the floor, facing down, tilted copies would break the closed form: kept parallel, offset from the view centre
        lP, lI = scenes.quad((0.3, -0.2, 1.5), (0.3, 1.8, 1.5), (1.3, 1.8, 1.5), (1.3, -0.2, 1.5))      # cross(p0-p2, p1-p2) points down (-z)
        emitter = ctx.create_material(scenes.LAMBERT, normal, ctx.solid_texture(7.0, 5.0, 3.0), color=black)
        ctx.create_instance([(ctx.create_mesh(lP, lI), emitter, True)])
        ctx.set_background(np.array([0, 0, 0, 1], np.float32), 1, 1)
    else:
        img = scenes.sky_sun_equirect()
        ctx.set_background(img, img.shape[1], img.shape[0])
    # a narrow lens straight down at the floor point (0.5, 0.4, 0): every pixel sees (nearly) the same irradiance
    lens = ctx.create_lens(ctx.make_lens(origin=(0.5, 0.4, 0.9), forward=(0, 0, -1), up=(0, 1, 0), vfov=0.02))
    return ctx.create_sensor(*extent), lens


def run_direct_lighting(make_ctx):
    rho = np.array([0.6, 0.5, 0.4])
    x0 = np.array([0.5, 0.4, 0.0])
    # (a) emissive quad: E = Le * integral over the quad of cos(theta_floor) cos(theta_light) / r^2 dA
    n = 2000
    ax = 0.3 + (np.arange(n) + 0.5) / n * 1.0; ay = -0.2 + (np.arange(n) + 0.5) / n * 2.0
    AX, AY = np.meshgrid(ax, ay, indexing="ij")
    d = np.stack([AX - x0[0], AY - x0[1], np.full_like(AX, 1.5)], -1)
    r2 = (d * d).sum(-1)
    form = ((1.5 * 1.5) / (r2 * r2)).sum() * (1.0 / n) * (2.0 / n)          # cos cos' / r^2 with cos = cos' = 1.5 / r
    want_mesh = rho / math.pi * np.array([7.0, 5.0, 3.0]) * form
    for nee in ((0, 1), (0, 0), (0, 4)):                                       # MIS with one light sample, no light sampling at all, four light samples
        c = make_ctx()
        s, l = direct_light_scene(c, "mesh")
        c.set_pipeline(samples_per_run=1, max_bounces=0, env_samples_per_bounce=nee[0], mesh_samples_per_bounce=nee[1])
        c.render(s, l, launches=2048 if nee[1] else 16384)
        got = c.sensor_data(s)[..., :3].astype(np.float64).reshape(-1, 3).mean(0)
        assert np.abs(got / want_mesh - 1).max() < (0.01 if nee[1] else 0.03), "mesh-light direct lighting, %d light samples: %s, quadrature %s" % (nee[1], got, want_mesh)
    # (b) environment: E = sum over the equal-area map's texels of L cos(theta)+ * 4 pi / S^2 (the map the renderer samples, built by its own pre-pass)
    c = make_ctx()
    s, l = direct_light_scene(c, "env")
    rgb, lum = c.env()
    S = rgb.shape[0]
    centres = (np.stack(np.meshgrid(np.arange(S), np.arange(S), indexing="xy"), -1).reshape(-1, 2) + 0.5) / S
    dirs = ss.square_to_equal_area_sphere(centres)
    L = np.asarray(rgb, np.float64)[..., :3].reshape(-1, 3)                    # texel (x, y) -> row y, column x: the same order as `centres`
    E = (L * np.maximum(dirs[:, 2:3], 0.0)).sum(0) * 4 * math.pi / (S * S)
    want_env = rho / math.pi * E
    for env_n in (1, 0):
        c = make_ctx()
        s, l = direct_light_scene(c, "env")
        c.set_pipeline(samples_per_run=1, max_bounces=0, env_samples_per_bounce=env_n, mesh_samples_per_bounce=0)
        c.render(s, l, launches=4096 if env_n else 16384)
        got = c.sensor_data(s)[..., :3].astype(np.float64).reshape(-1, 3).mean(0)
        # point lookups (NEE / eval) against the bilinear lookup of incomingRadiance, and a 2e-3-steradian sun on a 256^2 map: 3 %
        assert np.abs(got / want_env - 1).max() < 0.03, "environment direct lighting, %d env samples: %s, quadrature %s" % (env_n, got, want_env)


def test_oracle_direct_lighting_matches_quadrature(orc):
    run_direct_lighting(lambda: orc.Context(threads=usable_cores()))


@pytest.mark.gpu
def test_hip_direct_lighting_matches_quadrature(gpu_api):
    run_direct_lighting(lambda: gpu_api.Context())
