"""Second source for the WHOLE path: an independent float64 numpy path tracer written from the reference's HLSL — raygen / dispatchUV
(main.hlsl:43-95), Rng (random.hlsl), Camera::generateRay (camera.hlsl), PathTracingIntegrator::incomingRadiance (integrator.hlsl:68-183:
shading-frame choice, emission with MIS, the max-bounce / Russian-roulette rule, estimateDirectMISLight for the environment and the mesh
lights, material sampling, ray offsetting), EnvMap / MeshLights (light.hlsl), sampleAlias (mappings.hlsl:103-126) and the host's alias table
(alias_table.zig:25-92 over Accel.zig:503-519's triangle areas, in f32 as the host builds it) — on top of tests/second_source.py's shading math.

Test infrastructure, not product; it shares no code with oracle/ or moonshine_amd/.  Geometry is intersected by brute force (double-precision
Moller-Trumbore against every triangle), so it needs no BVH and runs scenes of a few hundred triangles at a few thousand paths.

Same seeds -> same random numbers (the PCG is integer-exact), so a path here takes the decisions the f32 implementations take unless a
decision (a coin flip, Russian roulette, a silhouette, a shadow ray grazing an edge) sits within f32 rounding of its threshold.  The tests
therefore compare PIXEL BY PIXEL at one sample per pixel and count the pixels that differ."""
import numpy as np

from tests import second_source as ss

MAX_UINT = 0xFFFFFFFF
RR_AFTER_BOUNCE = 3      # integrator.hlsl:130 `bounceCount > 3` (a module constant so that a test can show the comparison notices a different one)


# ---------------------------------------------------------------------------------------------------------------- random.hlsl
def _lcg(a):
    return a * np.uint32(747796405) + np.uint32(2891336453)


def _rxs_m_xs(a):
    b = ((a >> ((a >> np.uint32(28)) + np.uint32(4))) ^ a) * np.uint32(277803737)
    return (b >> np.uint32(22)) ^ b


def _pcg(a):
    return _rxs_m_xs(_lcg(a))


class Rng:
    def __init__(self, sx, sy, sz):
        with np.errstate(over="ignore"):
            u = lambda v: np.asarray(v, np.uint32)
            self.state = _pcg(u(sx) + _pcg(u(sy) + _pcg(u(sz))))

    def get(self, idx):
        """getFloat() for the paths idx -> float64 holding the exact f32 value"""
        with np.errstate(over="ignore"):
            self.state[idx] = _lcg(self.state[idx])
            return (_rxs_m_xs(self.state[idx]) >> np.uint32(8)).astype(np.float64) * 2.0 ** -24

    def get2(self, idx):
        a = self.get(idx); b = self.get(idx)      # float2(rng.getFloat(), rng.getFloat()): left to right
        return np.stack([a, b], -1)


# ---------------------------------------------------------------------------------------------------------------- alias_table.zig:25-92
def alias_table(weights):
    """Vose's method exactly as the host runs it, in f32 -> (alias u32[n], select f32[n], sum f32)"""
    f = np.float32
    w = np.asarray(weights, f); n = len(w)
    total = f(0.0)
    for x in w:
        total = f(total + x)
    alias = np.zeros(n, np.uint32); select = np.zeros(n, f)
    less = more = MAX_UINT
    for i in range(n):
        adj = f(f(w[i] * f(n)) / total)
        select[i] = adj
        if adj < f(1.0):
            alias[i] = less; less = i
        else:
            alias[i] = more; more = i
    while less != MAX_UINT and more != MAX_UINT:
        l = less; less = int(alias[l])
        m = more; more = int(alias[m])
        alias[l] = m
        select[m] = f(f(select[m] + select[l]) - f(1.0))
        if select[m] < f(1.0):
            alias[m] = less; less = m
        else:
            alias[m] = more; more = m
    while less != MAX_UINT:
        l = less; less = int(alias[l]); select[l] = f(1.0)
    return alias, select, total


# ---------------------------------------------------------------------------------------------------------------- the scene
class Scene:
    """spec: dict(textures=[HxWx4 float arrays], materials=[dict(type, normal, emissive, color, metalness, roughness, ior)],
    meshes=[dict(positions, indices, normals=None, texcoords=None)], instances=[dict(transform=3x4 or None, geometries=[(mesh, material, sampled)])],
    lens=dict(origin, forward, up, vfov, aperture, focus_distance), extent=(W, H), opts=dict(...)); env: second_source.EnvMap"""

    def __init__(self, spec, env):
        self.spec, self.env = spec, env
        o = dict(samples_per_run=1, max_bounces=4, env_samples_per_bounce=1, mesh_samples_per_bounce=1, flip_image=True, indexed_attributes=True, two_component_normal_texture=True)
        o.update(spec.get("opts", {})); self.opts = o
        f8 = np.float64
        P0, P1, P2, T0, T1, T2, N0, N1, N2, HT, HN, INST, MAT, SAMPLED, TW, TM, GEO, PRIM = ([] for _ in range(18))
        weights, wdata = [], []
        first_of = {}
        for ii, inst in enumerate(spec["instances"]):
            tw = np.eye(3, 4, dtype=np.float32) if inst.get("transform") is None else np.asarray(inst["transform"], np.float32).reshape(3, 4)
            a = tw[:, :3].astype(f8); ai = np.linalg.inv(a)
            tm = np.concatenate([ai, -(ai @ tw[:, 3:].astype(f8))], 1).astype(np.float32)      # the host inverts in f32 storage (Accel.zig:394-432)
            for gi, (mi, mat, sampled) in enumerate(inst["geometries"]):
                m = spec["meshes"][mi]
                pos = np.asarray(m["positions"], np.float32).astype(f8); idx = np.asarray(m["indices"], np.int64).reshape(-1, 3)
                nt = len(idx)
                att = idx if o["indexed_attributes"] else np.arange(3 * nt).reshape(nt, 3)        # world.hlsl:127-135
                first_of[(ii, gi)] = sum(len(x) for x in P0)
                P0.append(pos[idx[:, 0]]); P1.append(pos[idx[:, 1]]); P2.append(pos[idx[:, 2]])
                has_t, has_n = m.get("texcoords") is not None, m.get("normals") is not None
                tc = np.asarray(m["texcoords"], np.float32).astype(f8) if has_t else np.zeros((att.max() + 1, 2))
                nr = np.asarray(m["normals"], np.float32).astype(f8) if has_n else np.zeros((att.max() + 1, 3))
                T0.append(tc[att[:, 0]]); T1.append(tc[att[:, 1]]); T2.append(tc[att[:, 2]])
                N0.append(nr[att[:, 0]]); N1.append(nr[att[:, 1]]); N2.append(nr[att[:, 2]])
                HT.append(np.full(nt, has_t)); HN.append(np.full(nt, has_n))
                INST.append(np.full(nt, ii)); MAT.append(np.full(nt, mat)); SAMPLED.append(np.full(nt, bool(sampled)))
                GEO.append(np.full(nt, gi)); PRIM.append(np.arange(nt))
                TW.append(np.broadcast_to(tw.astype(f8), (nt, 3, 4))); TM.append(np.broadcast_to(tm.astype(f8), (nt, 3, 4)))
                if sampled:                                                                     # Accel.zig:503-519, f32
                    p32 = np.asarray(m["positions"], np.float32)
                    for k in range(nt):
                        q0, q1, q2 = (self._mul_point32(tw, p32[idx[k, j]]) for j in range(3))
                        c = self._cross32(q1 - q0, q2 - q0)
                        weights.append(np.float32(np.sqrt(np.float32(c[0] * c[0] + c[1] * c[1]) + np.float32(c[2] * c[2]), dtype=np.float32) / np.float32(2.0)))
                        wdata.append((ii, gi, k))
        cat = lambda L, shape: np.concatenate(L) if L else np.zeros(shape)
        self.p0, self.p1, self.p2 = cat(P0, (0, 3)), cat(P1, (0, 3)), cat(P2, (0, 3))
        self.t0, self.t1, self.t2 = cat(T0, (0, 2)), cat(T1, (0, 2)), cat(T2, (0, 2))
        self.n0, self.n1, self.n2 = cat(N0, (0, 3)), cat(N1, (0, 3)), cat(N2, (0, 3))
        self.has_t, self.has_n = cat(HT, (0,)).astype(bool), cat(HN, (0,)).astype(bool)
        self.inst, self.mat, self.sampled = cat(INST, (0,)).astype(int), cat(MAT, (0,)).astype(int), cat(SAMPLED, (0,)).astype(bool)
        self.tw, self.tm = cat(TW, (0, 3, 4)), cat(TM, (0, 3, 4))
        self.first_of = first_of
        # world-space triangles for the brute-force intersector
        w = lambda p: np.einsum("nij,nj->ni", self.tw[:, :, :3], p) + self.tw[:, :, 3]
        self.w0, self.w1, self.w2 = (w(p) for p in (self.p0, self.p1, self.p2)) if len(self.p0) else (self.p0, self.p1, self.p2)
        # mesh lights
        if weights:
            self.alias, self.select, self.alias_sum = alias_table(weights)
            self.alias_tri = np.array([first_of[(i, g)] + k for (i, g, k) in wdata], int)
        else:
            self.alias, self.select, self.alias_sum, self.alias_tri = np.zeros(0, np.uint32), np.zeros(0, np.float32), np.float32(0), np.zeros(0, int)
        self.textures = [np.asarray(t, np.float64) for t in spec["textures"]]

    @staticmethod
    def _mul_point32(m, p):   # vector.zig Mat3x4.mul_point in f32: row . (p, 1), summed left to right
        f = np.float32
        return np.array([f(f(f(m[r, 0] * p[0]) + f(m[r, 1] * p[1])) + f(m[r, 2] * p[2])) + m[r, 3] for r in range(3)], f)

    @staticmethod
    def _cross32(a, b):
        f = np.float32
        return np.array([f(a[1] * b[2]) - f(a[2] * b[1]), f(a[2] * b[0]) - f(a[0] * b[2]), f(a[0] * b[1]) - f(a[1] * b[0])], f)

    # ------------------------------------------------------------------ TraceRay stand-in: every triangle, double precision
    def _hits(self, o, d, chunk=4096):
        """-> t (inf where no hit), triangle, u, v for each ray (closest)"""
        n = len(o); T = np.full(n, np.inf); I = np.full(n, -1); U = np.zeros(n); V = np.zeros(n)
        if not len(self.w0):
            return T, I, U, V
        e1, e2 = self.w1 - self.w0, self.w2 - self.w0
        for a in range(0, n, chunk):
            oo, dd = o[a:a + chunk, None, :], d[a:a + chunk, None, :]
            with np.errstate(all="ignore"):
                pv = np.cross(dd, e2[None]); det = (e1[None] * pv).sum(-1)
                inv = 1.0 / det
                tv = oo - self.w0[None]
                u = (tv * pv).sum(-1) * inv
                qv = np.cross(tv, e1[None])
                v = (dd * qv).sum(-1) * inv
                t = (e2[None] * qv).sum(-1) * inv
            ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 0) & np.isfinite(t)
            t = np.where(ok, t, np.inf)
            k = t.argmin(1); r = np.arange(len(k))
            T[a:a + chunk] = t[r, k]; I[a:a + chunk] = np.where(np.isfinite(t[r, k]), k, -1); U[a:a + chunk] = u[r, k]; V[a:a + chunk] = v[r, k]
        return T, I, U, V

    def occluded(self, o, d, tmax):
        t, _, _, _ = self._hits(o, d)
        return t < tmax

    # ------------------------------------------------------------------ world.hlsl / material.hlsl lookups for triangles `tri` at barycentrics `uv`
    def attributes(self, tri, uv):
        return ss.mesh_attributes(self.p0[tri], self.p1[tri], self.p2[tri], self.t0[tri], self.t1[tri], self.t2[tri], self.n0[tri], self.n1[tri], self.n2[tri],
                                  uv, self.has_t[tri], self.has_n[tri], self.tw[tri], self.tm[tri])

    def sample_texture(self, which, mat, texcoord):
        """dTextures[materials[mat].<which>].SampleLevel(dTextureSampler, texcoord, 0) -> (n, 4)"""
        out = np.zeros((len(mat), 4))
        for m in np.unique(mat):
            sel = mat == m
            out[sel] = ss.vk_sample_linear(self.textures[self.spec["materials"][m][which]], texcoord[sel], mirrored=False)
        return out

    def mat_field(self, mat, key):
        return np.array([self.spec["materials"][m][key] for m in mat])


# ---------------------------------------------------------------------------------------------------------------- light.hlsl
def env_sample(sc, position, normal, rand):
    d, rad, pdf, _ = sc.env.sample(rand)
    test = pdf > 0
    if test.any():
        o = ss.offset_along_normal(position[test], ss.face_forward(normal[test], d[test])).astype(np.float64)
        occ = sc.occluded(o, d[test], np.inf)
        pdf = pdf.copy(); k = np.flatnonzero(test); pdf[k[occ]] = 0.0
    return d, rad, pdf


def mesh_sample(sc, position, normal, rand):
    n = len(position)
    count, total = len(sc.alias), float(sc.alias_sum)
    d, rad, pdf = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
    if count == 0 or total == 0 or n == 0:
        return d, rad, pdf
    scaled = rand[:, 0] * count                                   # sampleAlias (mappings.hlsl:115-126): rand.x is remapped IN PLACE
    idx = np.minimum(scaled.astype(np.int64), count - 1)
    rx = scaled - np.floor(scaled)
    take, rx = ss.coin_flip_remap(sc.select[idx].astype(np.float64), rx)
    idx = np.where(take, idx, sc.alias[idx].astype(np.int64))
    bary = ss.square_to_triangle(np.stack([rx, rand[:, 1]], -1))
    tri = sc.alias_tri[idx]
    pos, tc, tf, _ = sc.attributes(tri, bary)
    rad = sc.sample_texture("emissive", sc.mat[tri], tc)[:, :3]
    d = ss.normalize(pos - position)
    with np.errstate(all="ignore"):
        pdf = ss.area_to_solid_angle(pos, position, d, tf[0]) / total
    o_light = ss.offset_along_normal(pos, tf[0]).astype(np.float64)
    o_shade = ss.offset_along_normal(position, ss.face_forward(normal, d)).astype(np.float64)
    seg = o_light - o_shade
    tmax = np.sqrt(ss.dot(seg, seg))
    test = pdf > 0
    if test.any():
        occ = sc.occluded(o_shade[test], ss.normalize(seg[test]), tmax[test])
        k = np.flatnonzero(test); pdf = pdf.copy(); pdf[k[occ]] = 0.0
    return d, rad, pdf


# ---------------------------------------------------------------------------------------------------------------- material dispatch
def _material(types, color, metalness, roughness, ior, wi, wo, sq):
    n = len(types)
    out = dict(pdf=np.zeros(n), eval=np.zeros((n, 3)), dir=np.zeros((n, 3)), sample_pdf=np.zeros(n))
    for t in np.unique(types):
        s = types == t
        with np.errstate(all="ignore"):
            r = ss.material(int(t), color[s], metalness[s], roughness[s], ior[s], wi[s], wo[s], sq[s])
        for k in out:
            out[k][s] = r[k]
    return out


def _direct(sc, light, n_samples, frame, mat, out_fs, position, tri_n, rand):
    """estimateDirectMISLight (integrator.hlsl:20-36) -> rgb"""
    d, rad, pdf = light(sc, position, tri_n, rand)
    res = np.zeros((len(position), 3))
    ok = pdf > 0
    if not ok.any():
        return res
    l_fs = ss.world_to_frame(frame[0], frame[1], frame[2], d)
    m = _material(mat["type"], mat["color"], mat["metalness"], mat["roughness"], mat["ior"], l_fs, out_fs, np.zeros((len(position), 2)))
    ok &= m["pdf"] > 0
    with np.errstate(all="ignore"):
        w = ss.power_heuristic(n_samples, pdf, 1, m["pdf"])
        val = rad * m["eval"] * np.abs(l_fs[:, 2:3]) * (w / pdf)[:, None]
    res[ok] = val[ok]
    return res


# ---------------------------------------------------------------------------------------------------------------- integrator.hlsl:68-183
def incoming_radiance(sc, o, d, rng, stats=None):
    n = len(o); opt = sc.opts
    o, d = o.copy(), d.copy()
    acc, thr = np.zeros((n, 3)), np.ones((n, 3))
    bounce = np.zeros(n, int); last_pdf = np.zeros(n); last_delta = np.zeros(n, bool); alive = np.ones(n, bool)
    env_n, mesh_n, max_b = opt["env_samples_per_bounce"], opt["mesh_samples_per_bounce"], opt["max_bounces"]
    while alive.any():
        idx = np.flatnonzero(alive)
        t, tri, u, v = sc._hits(o[idx], d[idx])
        miss = tri < 0
        if miss.any():                                                              # :163-180
            k = idx[miss]
            plain = (env_n == 0) | (bounce[k] == 0) | last_delta[k]
            uv = ss.square_to_equal_area_sphere_inverse(d[k])
            rad_plain = ss.vk_sample_linear(sc.env.rgb, uv, mirrored=True)            # EnvMap::incomingRadiance (light.hlsl:99-102)
            rad_e, pdf_e = sc.env.eval(d[k])
            with np.errstate(all="ignore"):
                w = ss.power_heuristic(1, last_pdf[k], env_n, pdf_e)
            add = np.where(plain[:, None], rad_plain, np.where((pdf_e > 0)[:, None], rad_e * w[:, None], 0.0))
            acc[k] += thr[k] * add
            alive[k] = False
        idx, tri, u, v = idx[~miss], tri[~miss], u[~miss], v[~miss]
        if not len(idx):
            break
        pos, tc, tf, fr = sc.attributes(tri, np.stack([u, v], -1))
        mat = sc.mat[tri]
        texel = sc.sample_texture("normal", mat, tc)
        xf = ss.texture_frame(texel[:, :3], fr[0], fr[1], fr[2], np.full(len(idx), bool(opt["two_component_normal_texture"])))
        emissive = sc.sample_texture("emissive", mat, tc)[:, :3]
        types = sc.mat_field(mat, "type")
        M = dict(type=types, color=sc.sample_texture("color", mat, tc)[:, :3], metalness=sc.sample_texture("metalness", mat, tc)[:, 0],
                 roughness=sc.sample_texture("roughness", mat, tc)[:, 0], ior=sc.mat_field(mat, "ior").astype(np.float64))
        out_ws = -d[idx]
        front = ss.dot(tf[0], out_ws) > 0                                            # :91-103
        def valid(nrm):
            c = ss.dot(out_ws, nrm)
            return (front & (c > 0)) | (~front & (-c > 0))
        use_x, use_f = valid(xf[0]), valid(fr[0])
        pick = lambda a, b, c_: np.where(use_x[:, None], a, np.where(use_f[:, None], b, c_))
        sf = tuple(pick(xf[j], fr[j], tf[j]) for j in range(3))
        out_ss = ss.world_to_frame(sf[0], sf[1], sf[2], out_ws)
        # emission (:108-124)
        sampled = sc.sampled[tri]
        unsampled_rule = (mesh_n == 0) | (bounce[idx] == 0) | ~sampled | last_delta[idx]
        facing = ss.dot(out_ws, tf[0]) > 0
        with np.errstate(all="ignore"):
            lpdf = ss.area_to_solid_angle(pos, o[idx], d[idx], tf[0]) / float(sc.alias_sum) if len(sc.alias) else np.zeros(len(idx))
            w = ss.power_heuristic(1, last_pdf[idx], mesh_n, lpdf)
        add = np.where((unsampled_rule & facing)[:, None], emissive, np.where((~unsampled_rule & sampled & (lpdf > 0))[:, None], emissive * w[:, None], 0.0))
        acc[idx] += thr[idx] * add
        # termination (:128-135)
        stop = bounce[idx] >= max_b + 1
        rr = ~stop & (bounce[idx] > RR_AFTER_BOUNCE)
        if rr.any():
            k = idx[rr]
            p = np.minimum(0.95, ss.luminance(thr[k]))
            lost = rng.get(k) > p
            with np.errstate(all="ignore"):
                thr[k] = thr[k] / p[:, None]
            stop[np.flatnonzero(rr)[lost]] = True
        alive[idx[stop]] = False
        keep = ~stop
        idx, tri, pos = idx[keep], tri[keep], pos[keep]
        if not len(idx):
            continue
        sf = tuple(x[keep] for x in sf); tfn = tf[0][keep]; out_ss = out_ss[keep]
        M = {k_: v_[keep] for k_, v_ in M.items()}
        delta = (M["type"] == ss.PERFECT_MIRROR) | (M["type"] == ss.GLASS)
        nd = np.flatnonzero(~delta)
        if len(nd):                                                                 # :139-151
            sub = lambda x: x[nd]
            Mn = {k_: v_[nd] for k_, v_ in M.items()}; sfn = tuple(x[nd] for x in sf)
            for _ in range(env_n):
                rand = rng.get2(idx[nd])
                acc[idx[nd]] += thr[idx[nd]] * _direct(sc, env_sample, env_n, sfn, Mn, sub(out_ss), sub(pos), sub(tfn), rand) / env_n
            for _ in range(mesh_n):
                rand = rng.get2(idx[nd])
                acc[idx[nd]] += thr[idx[nd]] * _direct(sc, mesh_sample, mesh_n, sfn, Mn, sub(out_ss), sub(pos), sub(tfn), rand) / mesh_n
        sq = rng.get2(idx)                                                           # :154-165
        s = _material(M["type"], M["color"], M["metalness"], M["roughness"], M["ior"], np.zeros((len(idx), 3)) + [0, 0, 1.0], out_ss, sq)
        dead = s["sample_pdf"] == 0                                                  # (a NaN pdf compares false with 0 in HLSL too: the path goes on)
        e = _material(M["type"], M["color"], M["metalness"], M["roughness"], M["ior"], s["dir"], out_ss, sq)["eval"]
        new_d = ss.frame_to_world(sf[0], sf[1], sf[2], s["dir"])
        new_o = ss.offset_along_normal(pos, ss.face_forward(tfn, new_d)).astype(np.float64)
        with np.errstate(all="ignore"):
            f = e * (np.abs(s["dir"][:, 2]) / s["sample_pdf"])[:, None]
        live = ~dead; k = idx[live]
        last_pdf[k] = s["sample_pdf"][live]; d[k] = new_d[live]; o[k] = new_o[live]; thr[k] = thr[k] * f[live]
        bounce[k] += 1; last_delta[k] = delta[live]
        alive[idx[dead]] = False
        if stats is not None:
            stats["bounces"] = stats.get("bounces", 0) + 1
    return acc


# ---------------------------------------------------------------------------------------------------------------- main.hlsl:43-95
def render_launch(sc, sample_index=0):
    """one launch at samples_per_run = 1 -> (H, W, 3) radiance of that sample (storeColor with sampleCount = 0 stores it as is)"""
    W, H = sc.spec["extent"]; opt = sc.opts
    assert opt["samples_per_run"] == 1
    ys, xs = np.mgrid[0:H, 0:W]; xs, ys = xs.ravel(), ys.ravel()
    rng = Rng(np.full(len(xs), sample_index), xs, ys)
    every = np.arange(len(xs))
    r1 = rng.get2(every)
    centre = 0.5 + 0.5 * ss.square_to_gaussian(r1)
    uv = (np.stack([xs, ys], -1) + centre) / np.array([W, H], np.float64)
    if opt["flip_image"]:
        uv[:, 1] = 1.0 - uv[:, 1]
    r2 = rng.get2(every)
    L = sc.spec["lens"]; b = lambda v: np.broadcast_to(np.asarray(v, np.float32).astype(np.float64), (len(xs),) + np.shape(v)).copy()
    o, d = ss.camera_generate_ray(b(L["origin"]), b(L["forward"]), b(L["up"]), b(L["vfov"]), b(L["aperture"]), b(L["focus_distance"]), float(W), float(H), uv, r2)
    return incoming_radiance(sc, o, d, rng).reshape(H, W, 3)


# ---------------------------------------------------------------------------------------------------------------- the same spec into a context
def build_context(ctx, spec):
    """pushes the spec through the C ABI of a context (moonshine_amd.api.Context or the oracle's) -> (sensor, lens)"""
    tex = [ctx.create_texture(np.ascontiguousarray(t, np.float32), t.shape[1], t.shape[0], "r32g32b32a32_sfloat") for t in spec["textures"]]
    mats = [ctx.create_material(m["type"], tex[m["normal"]], tex[m["emissive"]], color=tex[m["color"]], metalness=tex[m["metalness"]],
                                roughness=tex[m["roughness"]], ior=m["ior"]) for m in spec["materials"]]
    meshes = [ctx.create_mesh(np.asarray(m["positions"], np.float32), np.asarray(m["indices"], np.uint32), normals=m.get("normals"), texcoords=m.get("texcoords")) for m in spec["meshes"]]
    for inst in spec["instances"]:
        ctx.create_instance([(meshes[a], mats[b], bool(c)) for (a, b, c) in inst["geometries"]], transform=inst.get("transform"))
    o = dict(samples_per_run=1, max_bounces=4, env_samples_per_bounce=1, mesh_samples_per_bounce=1); o.update(spec.get("opts", {}))
    ctx.set_pipeline(**o)
    if spec.get("background") is not None:
        img = spec["background"]; ctx.set_background(img, img.shape[1], img.shape[0])
    L = spec["lens"]
    lens = ctx.create_lens(ctx.make_lens(tuple(L["origin"]), tuple(L["forward"]), tuple(L["up"]), L["vfov"], L["aperture"], L["focus_distance"]))
    return ctx.create_sensor(*spec["extent"]), lens
