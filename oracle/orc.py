"""ORACLE — test infrastructure only.

ctypes binding of oracle/liborc.so (the scalar C restatement of the reference's hrtsystem hot
path).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The method names mirror moonshine_amd.api.Context so one scene script can drive both.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class F32x2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class F32x3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class F32x4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Mat3x4(C.Structure):
    _fields_ = [("x", F32x4), ("y", F32x4), ("z", F32x4)]


class Geometry(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("material", C.c_uint32), ("sampled", C.c_bool)]


class Extent2D(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32)]


class Lens(C.Structure):
    _fields_ = [("origin", F32x3), ("forward", F32x3), ("up", F32x3),
                ("vfov", C.c_float), ("aperture", C.c_float), ("focus_distance", C.c_float)]


class MsneMaterialDesc(C.Structure):
    _fields_ = [("normal", C.c_uint32), ("emissive", C.c_uint32), ("type", C.c_uint32),
                ("color", C.c_uint32), ("metalness", C.c_uint32), ("roughness", C.c_uint32), ("ior", C.c_float)]


class MsnePipelineOpts(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("samples_per_run", "max_bounces", "env_samples_per_bounce",
                                           "mesh_samples_per_bounce", "flip_image", "indexed_attributes",
                                           "two_component_normal_texture")]


class MsneConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("tile_size", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32)]


GLASS, LAMBERT, PERFECT_MIRROR, STANDARD_PBR = 0, 1, 2, 3
FORMATS = {"r8g8b8a8_srgb": 0, "r8g8_unorm": 1, "r8_unorm": 2, "r32g32b32a32_sfloat": 3,
           "r32g32_sfloat": 4, "r32_sfloat": 5, "r16g16b16a16_sfloat": 6}


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        vp, u32, i64, f32p = C.c_void_p, C.c_uint32, C.c_int64, C.POINTER(C.c_float)
        L.OrcCreate.restype = vp; L.OrcCreate.argtypes = [C.POINTER(MsneConfig)]
        L.OrcDestroy.argtypes = [vp]
        L.OrcSetThreads.argtypes = [vp, C.c_int]
        L.OrcCreateMesh.restype = i64
        L.OrcCreateMesh.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t]
        L.OrcCreateTexture.restype = i64; L.OrcCreateTexture.argtypes = [vp, vp, Extent2D, C.c_int]
        L.OrcCreateSolidTexture1.restype = i64; L.OrcCreateSolidTexture1.argtypes = [vp, C.c_float]
        L.OrcCreateSolidTexture2.restype = i64; L.OrcCreateSolidTexture2.argtypes = [vp, F32x2]
        L.OrcCreateSolidTexture3.restype = i64; L.OrcCreateSolidTexture3.argtypes = [vp, F32x3]
        L.OrcCreateMaterial.restype = i64; L.OrcCreateMaterial.argtypes = [vp, C.POINTER(MsneMaterialDesc)]
        L.OrcSetMaterial.argtypes = [vp, u32, C.POINTER(MsneMaterialDesc)]
        L.OrcCreateInstance.restype = i64; L.OrcCreateInstance.argtypes = [vp, Mat3x4, C.POINTER(Geometry), C.c_size_t, C.c_bool]
        L.OrcSetInstanceTransform.argtypes = [vp, u32, Mat3x4]
        L.OrcSetInstanceVisibility.argtypes = [vp, u32, C.c_bool]
        L.OrcSetGeometryMaterial.restype = C.c_int; L.OrcSetGeometryMaterial.argtypes = [vp, u32, u32, u32]
        L.OrcSetExhaustiveSearch.restype = None; L.OrcSetExhaustiveSearch.argtypes = [vp, C.c_int]
        L.OrcSetPipeline.argtypes = [vp, C.POINTER(MsnePipelineOpts)]
        L.OrcSetBackground.argtypes = [vp, vp, Extent2D]
        L.OrcCreateSensor.restype = i64; L.OrcCreateSensor.argtypes = [vp, Extent2D]
        L.OrcGetSensorData.restype = f32p; L.OrcGetSensorData.argtypes = [vp, u32]
        L.OrcGetSampleCount.restype = u32; L.OrcGetSampleCount.argtypes = [vp, u32]
        L.OrcClearSensor.argtypes = [vp, u32]
        L.OrcCreateLens.restype = i64; L.OrcCreateLens.argtypes = [vp, Lens]
        L.OrcSetLens.argtypes = [vp, u32, Lens]
        L.OrcRender.argtypes = [vp, u32, u32, u32]
        L.OrcGetCounters.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.OrcResetCounters.argtypes = [vp]
        L.OrcTraceClosest.argtypes = [vp, vp, vp, C.c_float, vp, vp]
        L.OrcTraceShadow.argtypes = [vp, vp, vp, C.c_float]
        L.OrcGenerateRay.argtypes = [C.POINTER(Lens), u32, u32, C.c_float, C.c_float, C.c_float, C.c_float, vp]
        L.OrcEnvSize.restype = u32; L.OrcEnvSize.argtypes = [vp]
        L.OrcEnvRgb.restype = f32p; L.OrcEnvRgb.argtypes = [vp]
        L.OrcEnvLum.restype = f32p; L.OrcEnvLum.argtypes = [vp, u32]
        L.OrcAliasCount.restype = u32; L.OrcAliasCount.argtypes = [vp]
        L.OrcAliasTable.restype = vp; L.OrcAliasTable.argtypes = [vp]
        L.OrcWorldToInstance.argtypes = [vp, u32, vp]
        L.OrcMathProbe.argtypes = [C.c_int, vp, vp, u32]
        L.OrcSquareToEqualAreaSphere.argtypes = [vp, vp, u32]
        L.OrcSquareToEqualAreaSphereInverse.argtypes = [vp, vp, u32]
        L.OrcOffsetAlongNormal.argtypes = [vp, vp, vp, u32]
        L.OrcRngFloats.argtypes = [u32, u32, u32, vp, u32, vp]
        L.OrcPcg.restype = u32; L.OrcPcg.argtypes = [u32]
        L.OrcBuildAliasTable.argtypes = [vp, u32, vp, vp, vp]
        L.OrcBsdfProbe.argtypes = [u32, vp, vp, vp, vp, vp]
        L.OrcProbeBatch.restype = C.c_int; L.OrcProbeBatch.argtypes = [vp, C.c_int, vp, u32, vp]
        L.OrcDebugPath.restype = u32; L.OrcDebugPath.argtypes = [vp, u32, u32, u32, u32, u32, vp, vp, u32, C.POINTER(C.c_uint64)]
        _LIB = L
    return _LIB


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def mat3x4(rows=None):
    m = Mat3x4()
    r = np.eye(3, 4, dtype=np.float32) if rows is None else _f32(rows, (3, 4))
    C.memmove(C.byref(m), r.ctypes.data, 48)
    return m


def make_lens(origin, forward, up, vfov, aperture=0.0, focus_distance=1.0):
    return Lens(F32x3(*origin), F32x3(*forward), F32x3(*up), vfov, aperture, focus_distance)


class Context:
    """Same surface as moonshine_amd.api.Context (subset needed by tests/bench)."""

    def __init__(self, tile_size=64, shard_index=0, shard_count=1, threads=1):
        self.L = lib()
        cfg = MsneConfig(-1, tile_size, shard_index, shard_count)
        self.h = self.L.OrcCreate(C.byref(cfg))
        self.L.OrcSetThreads(self.h, threads)
        self._extents = {}

    make_lens = staticmethod(make_lens)

    def close(self):
        if self.h:
            self.L.OrcDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def create_mesh(self, positions, indices, normals=None, texcoords=None):
        p = _f32(positions, (-1, 3)); i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        n = _f32(normals, (-1, 3)) if normals is not None else None
        t = _f32(texcoords, (-1, 2)) if texcoords is not None else None
        ac = len(n) if n is not None else (len(t) if t is not None else 0)
        return int(self.L.OrcCreateMesh(self.h, _ptr(p), _ptr(n) if n is not None else None,
                                        _ptr(t) if t is not None else None, len(p), ac, _ptr(i), len(i)))

    def solid_texture(self, *v):
        if len(v) == 1:
            return int(self.L.OrcCreateSolidTexture1(self.h, v[0]))
        if len(v) == 2:
            return int(self.L.OrcCreateSolidTexture2(self.h, F32x2(*v)))
        return int(self.L.OrcCreateSolidTexture3(self.h, F32x3(*v)))

    def create_texture(self, data, width, height, fmt):
        a = np.ascontiguousarray(data)
        return int(self.L.OrcCreateTexture(self.h, _ptr(a), Extent2D(width, height), FORMATS[fmt]))

    def create_material(self, type, normal, emissive, color=0, metalness=0, roughness=0, ior=1.5):
        d = MsneMaterialDesc(normal, emissive, type, color, metalness, roughness, ior)
        return int(self.L.OrcCreateMaterial(self.h, C.byref(d)))

    def create_instance(self, geometries, transform=None, visible=True):
        arr = (Geometry * len(geometries))(*[Geometry(m, mat, bool(s)) for (m, mat, s) in geometries])
        return int(self.L.OrcCreateInstance(self.h, mat3x4(transform), arr, len(geometries), visible))

    def set_instance_transform(self, h, transform):
        self.L.OrcSetInstanceTransform(self.h, h, mat3x4(transform))

    def set_instance_visibility(self, h, v):
        self.L.OrcSetInstanceVisibility(self.h, h, v)

    def set_exhaustive_search(self, level):
        """0: the instances' and the triangles' boxes cull (what every other test runs); 1: every visible instance is entered; 2: and every one of its triangles tested"""
        self.L.OrcSetExhaustiveSearch(self.h, int(level))

    def set_geometry_material(self, instance, geometry_index, material):
        if self.L.OrcSetGeometryMaterial(self.h, instance, geometry_index, material) != 0:
            raise RuntimeError("OrcSetGeometryMaterial: unknown instance, geometry or material")

    def set_pipeline(self, samples_per_run=1, max_bounces=4, env_samples_per_bounce=1, mesh_samples_per_bounce=1,
                     flip_image=True, indexed_attributes=True, two_component_normal_texture=True):
        o = MsnePipelineOpts(samples_per_run, max_bounces, env_samples_per_bounce, mesh_samples_per_bounce,
                             int(flip_image), int(indexed_attributes), int(two_component_normal_texture))
        self.L.OrcSetPipeline(self.h, C.byref(o))

    def set_background(self, rgba, width, height):
        a = _f32(rgba, (height, width, 4))
        assert self.L.OrcSetBackground(self.h, _ptr(a), Extent2D(width, height)) == 0

    def create_sensor(self, width, height):
        s = int(self.L.OrcCreateSensor(self.h, Extent2D(width, height)))
        self._extents[s] = (width, height)
        return s

    def create_lens(self, lens):
        return int(self.L.OrcCreateLens(self.h, lens))

    def set_lens(self, h, lens):
        self.L.OrcSetLens(self.h, h, lens)

    def render(self, sensor, lens, launches=1):
        assert self.L.OrcRender(self.h, sensor, lens, launches) == 0

    def clear_sensor(self, sensor):
        self.L.OrcClearSensor(self.h, sensor)

    def sample_count(self, sensor):
        return int(self.L.OrcGetSampleCount(self.h, sensor))

    def sensor_data(self, sensor):
        w, h = self._extents[sensor]
        p = self.L.OrcGetSensorData(self.h, sensor)
        return np.ctypeslib.as_array(p, shape=(h, w, 4)).copy()

    def debug_path(self, sensor, lens, k, x, y, cap=64):
        """one camera path (sample index k of pixel x, y) -> (radiance (3,), records (n, 12): instance, primitive, t, u, v, direction, origin, geometry of every
        surface hit, (closest rays, shadow rays) of the path)"""
        rgb = np.zeros(3, np.float32); rec = np.zeros((cap, 12), np.float32); cnt = (C.c_uint64 * 2)()
        n = self.L.OrcDebugPath(self.h, sensor, lens, k, x, y, _ptr(rgb), _ptr(rec), cap, cnt)
        return rgb, rec[:n], (int(cnt[0]), int(cnt[1]))

    def counters(self):
        out = (C.c_uint64 * 8)()
        self.L.OrcGetCounters(self.h, out)
        k = ("closest_rays", "shadow_rays", "samples", "surface_hits", "node_visits", "tri_tests", "shadow_node_visits", "shadow_tri_tests")
        return dict(zip(k, [int(x) for x in out]))

    def reset_counters(self):
        self.L.OrcResetCounters(self.h)

    def trace_closest(self, o, d, tmax=1e12):
        ids = np.zeros(3, np.uint32); tuv = np.zeros(3, np.float32)
        r = self.L.OrcTraceClosest(self.h, _ptr(_f32(o)), _ptr(_f32(d)), tmax, _ptr(ids), _ptr(tuv))
        return bool(r), ids, tuv

    def trace_shadow(self, o, d, tmax=1e12):
        return bool(self.L.OrcTraceShadow(self.h, _ptr(_f32(o)), _ptr(_f32(d)), tmax))

    def env(self):
        s = int(self.L.OrcEnvSize(self.h))
        rgb = np.ctypeslib.as_array(self.L.OrcEnvRgb(self.h), shape=(s, s, 4)).copy()
        lum = []
        l = 0
        while (s >> l) >= 1:
            d = s >> l
            lum.append(np.ctypeslib.as_array(self.L.OrcEnvLum(self.h, l), shape=(d, d)).copy())
            if d == 1:
                break
            l += 1
        return rgb, lum

    def alias_table(self):
        n = int(self.L.OrcAliasCount(self.h))
        dt = np.dtype([("alias", "<u4"), ("select", "<f4"), ("instance", "<u4"), ("geometry", "<u4"), ("primitive", "<u4")])
        buf = C.string_at(self.L.OrcAliasTable(self.h), n * dt.itemsize)
        return np.frombuffer(buf, dtype=dt).copy()

    def world_to_instance(self, inst):
        out = np.zeros(12, np.float32)
        self.L.OrcWorldToInstance(self.h, inst, _ptr(out))
        return out.reshape(3, 4)


# ---- stateless probes ----
def math_probe(fn, x):
    names = {"sin": 0, "cos": 1, "log": 2, "acos": 3, "tan": 4, "atan2": 5}
    x = _f32(x)
    n = x.size // 2 if fn == "atan2" else x.size
    out = np.zeros(n, np.float32)
    lib().OrcMathProbe(names[fn], _ptr(x), _ptr(out), n)
    return out


def rng_floats(s, x, y, n):
    out = np.zeros(n, np.float32); st = C.c_uint32()
    lib().OrcRngFloats(s, x, y, _ptr(out), n, C.byref(st))
    return st.value, out


def pcg(a):
    return int(lib().OrcPcg(a))


def square_to_equal_area_sphere(uv):
    uv = _f32(uv, (-1, 2)); out = np.zeros((len(uv), 3), np.float32)
    lib().OrcSquareToEqualAreaSphere(_ptr(uv), _ptr(out), len(uv)); return out


def square_to_equal_area_sphere_inverse(d):
    d = _f32(d, (-1, 3)); out = np.zeros((len(d), 2), np.float32)
    lib().OrcSquareToEqualAreaSphereInverse(_ptr(d), _ptr(out), len(d)); return out


def offset_along_normal(p, n):
    p = _f32(p, (-1, 3)); n = _f32(n, (-1, 3)); out = np.zeros_like(p)
    lib().OrcOffsetAlongNormal(_ptr(p), _ptr(n), _ptr(out), len(p)); return out


def build_alias_table(weights):
    w = _f32(weights); n = len(w)
    alias = np.zeros(n, np.uint32); select = np.zeros(n, np.float32); s = C.c_float()
    lib().OrcBuildAliasTable(_ptr(w), n, _ptr(alias), _ptr(select), C.byref(s))
    return alias, select, s.value


def generate_ray(lens, W, H, u, v, r0=0.5, r1=0.5):
    out = np.zeros(6, np.float32)
    lib().OrcGenerateRay(C.byref(lens), W, H, u, v, r0, r1, _ptr(out))
    return out


def bsdf_probe(type, color, metalness, roughness, ior, wi, wo, sq):
    params = _f32(list(color) + [metalness, roughness, ior]); out = np.zeros(8, np.float32)
    lib().OrcBsdfProbe(type, _ptr(params), _ptr(_f32(wi)), _ptr(_f32(wo)), _ptr(_f32(sq)), _ptr(out))
    return {"pdf": float(out[0]), "eval": out[1:4].copy(), "dir": out[4:7].copy(), "sample_pdf": float(out[7])}


# batch probes (oracle/orc_core.c OrcProbeBatch; the product mirrors the table in MsneShadeProbe)
PROBES = {"bsdf": (0, 15, 8), "env_sample": (1, 2, 7), "env_eval": (2, 3, 4), "env_incoming": (3, 3, 3), "equal_area": (4, 2, 3),
          "equal_area_inverse": (5, 3, 2), "triangle": (6, 2, 2), "gaussian": (7, 2, 2), "cosine_hemisphere": (8, 2, 3),
          "fresnel_dielectric": (9, 3, 1), "offset_along_normal": (10, 6, 3), "coordinate_system": (11, 3, 6),
          "area_to_solid_angle": (12, 12, 1), "ggx": (13, 7, 3), "refract": (14, 7, 3), "power_heuristic": (15, 4, 1), "frame": (16, 9, 6), "texture": (17, 3, 4),
          "mesh_attributes": (18, 51, 23), "texture_frame": (19, 13, 9), "camera": (20, 18, 6)}


def probe(name, x, ctx=None):
    """x: (n, in_width) float32 -> (n, out_width) float32.  env_* probes need a context (its environment map)."""
    fn, wi, wo = PROBES[name]
    x = _f32(x, (-1, wi)); out = np.zeros((len(x), wo), np.float32)
    if lib().OrcProbeBatch(ctx.h if ctx is not None else None, fn, _ptr(x), len(x), _ptr(out)) != 0:
        raise RuntimeError("OrcProbeBatch(%s) failed" % name)
    return out
