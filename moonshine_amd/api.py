"""ctypes binding of libmoonshine_amd.so (include/moonshine_amd.h) — plumbing only.

`Context` mirrors the reference's object model for the hot path (engine/hrtsystem Scene / World /
Camera + hydra/hydra.zig's C ABI): meshes, textures, materials, instances, background, lenses,
sensors, pipeline constants, render.  There is NO CPU fallback: if the HIP library is missing or
no GPU is present, construction raises.
"""
import ctypes as C
import os

import numpy as np
from . import tiles

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSNE_LIB") or os.path.join(_HERE, "libmoonshine_amd.so")   # $MSNE_LIB: an experimental build (tools/variant_rates.py)
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


class Material(C.Structure):
    _fields_ = [("normal", C.c_uint32), ("emissive", C.c_uint32), ("color", C.c_uint32),
                ("metalness", C.c_uint32), ("roughness", C.c_uint32), ("ior", C.c_float)]


class MsneMaterialDesc(C.Structure):
    _fields_ = [("normal", C.c_uint32), ("emissive", C.c_uint32), ("type", C.c_uint32),
                ("color", C.c_uint32), ("metalness", C.c_uint32), ("roughness", C.c_uint32), ("ior", C.c_float)]


class MsnePipelineOpts(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("samples_per_run", "max_bounces", "env_samples_per_bounce",
                                           "mesh_samples_per_bounce", "flip_image", "indexed_attributes",
                                           "two_component_normal_texture")]


class MsneConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("tile_size", C.c_uint32), ("shard_index", C.c_uint32), ("shard_count", C.c_uint32)]


class MsneClickData(C.Structure):   # input.hlsl:24-29
    _fields_ = [("instance_index", C.c_int32), ("geometry_index", C.c_uint32), ("primitive_index", C.c_uint32), ("barycentrics", F32x2)]


class MsneStats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("samples", C.c_uint64), ("launches", C.c_uint64),
                ("trace_closest_ms", C.c_double), ("trace_shadow_ms", C.c_double), ("shade_ms", C.c_double), ("render_ms", C.c_double),
                ("trace_closest_launches", C.c_uint64), ("trace_shadow_launches", C.c_uint64), ("shade_launches", C.c_uint64)]


PRESENT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.c_uint32)   # MsnePresentFn

GLASS, LAMBERT, PERFECT_MIRROR, STANDARD_PBR = 0, 1, 2, 3
FORMATS = {"r8g8b8a8_srgb": 0, "r8g8_unorm": 1, "r8_unorm": 2, "r32g32b32a32_sfloat": 3,
           "r32g32_sfloat": 4, "r32_sfloat": 5, "r16g16b16a16_sfloat": 6}

# every symbol include/moonshine_amd.h declares: (name, restype, argtypes)
_vp, _u32, _i64 = C.c_void_p, C.c_uint32, C.c_int64
SYMBOLS = [
    ("HdMoonshineCreate", _vp, []),
    ("HdMoonshineDestroy", None, [_vp]),
    ("HdMoonshineRender", C.c_bool, [_vp, _u32, _u32]),
    ("HdMoonshineRebuildPipeline", C.c_bool, [_vp]),
    ("HdMoonshineCreateMesh", _u32, [_vp, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t]),
    ("HdMoonshineCreateSolidTexture1", _u32, [_vp, C.c_float, C.c_char_p]),
    ("HdMoonshineCreateSolidTexture2", _u32, [_vp, F32x2, C.c_char_p]),
    ("HdMoonshineCreateSolidTexture3", _u32, [_vp, F32x3, C.c_char_p]),
    ("HdMoonshineCreateRawTexture", _u32, [_vp, _vp, Extent2D, C.c_int, C.c_char_p]),
    ("HdMoonshineCreateMaterial", _u32, [_vp, Material]),
    ("HdMoonshineSetMaterialNormal", None, [_vp, _u32, _u32]),
    ("HdMoonshineSetMaterialEmissive", None, [_vp, _u32, _u32]),
    ("HdMoonshineSetMaterialColor", None, [_vp, _u32, _u32]),
    ("HdMoonshineSetMaterialMetalness", None, [_vp, _u32, _u32]),
    ("HdMoonshineSetMaterialRoughness", None, [_vp, _u32, _u32]),
    ("HdMoonshineSetMaterialIOR", None, [_vp, _u32, C.c_float]),
    ("HdMoonshineCreateInstance", _u32, [_vp, Mat3x4, C.POINTER(Geometry), C.c_size_t, C.c_bool]),
    ("HdMoonshineDestroyInstance", None, [_vp, _u32]),
    ("HdMoonshineSetInstanceTransform", None, [_vp, _u32, Mat3x4]),
    ("HdMoonshineSetInstanceVisibility", None, [_vp, _u32, C.c_bool]),
    ("HdMoonshineCreateSensor", _u32, [_vp, Extent2D]),
    ("HdMoonshineGetSensorData", C.POINTER(C.c_float), [_vp, _u32]),
    ("HdMoonshineCreateLens", _u32, [_vp, Lens]),
    ("HdMoonshineSetLens", None, [_vp, _u32, Lens]),
    ("MsneCreate", _vp, [C.POINTER(MsneConfig)]),
    ("MsneCreateMesh", _i64, [_vp, _vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp, C.c_size_t]),
    ("MsneCreateTexture", _i64, [_vp, _vp, Extent2D, C.c_int]),
    ("MsneCreateMaterial", _i64, [_vp, C.POINTER(MsneMaterialDesc)]),
    ("MsneSetGeometryMaterial", C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("MsneProbeClockGhz", C.c_double, [C.c_int]),
    ("MsneSetPipeline", C.c_int, [_vp, C.POINTER(MsnePipelineOpts)]),
    ("MsneGetPipeline", C.c_int, [_vp, C.POINTER(MsnePipelineOpts)]),
    ("MsneSetBackground", C.c_int, [_vp, _vp, Extent2D]),
    ("MsneRender", C.c_int, [_vp, _u32, _u32, _u32, C.c_int]),
    ("MsneReserve", C.c_int, [_vp, _u32, _u32]),
    ("MsneClearSensor", None, [_vp, _u32]),
    ("MsneGetSampleCount", _u32, [_vp, _u32]),
    ("MsneGetShardTileCount", C.c_uint64, [_vp, _u32]),
    ("MsneGetPackedFilmDevicePtr", _vp, [_vp, _u32]),
    ("MsneUnpackGatheredFilm", C.c_int, [_vp, _u32, _vp, _u32]),
    ("MsneGetStats", C.c_int, [_vp, C.POINTER(MsneStats)]),
    ("MsneResetStats", None, [_vp]),
    ("MsneGetLastError", C.c_char_p, [_vp]),
    ("MsneLoadGlb", C.c_int, [_vp, C.c_char_p, _vp]),
    ("MsneSetBackgroundExr", C.c_int, [_vp, C.c_char_p]),
    ("MsneSaveSensorExr", C.c_int, [_vp, _u32, Extent2D, C.c_char_p]),
    ("MsneExrLoad", C.c_int, [C.c_char_p, _vp, C.POINTER(Extent2D)]),
    ("MsneExrSave", C.c_int, [C.c_char_p, _vp, Extent2D]),
    ("MsneGetIoError", C.c_char_p, []),
    ("MsneSetProfiling", None, [_vp, C.c_int, C.c_int]),
    ("MsneGetTexelPoolBytes", C.c_uint64, [_vp]),
    ("MsneGetAccelStats", None, [_vp, C.POINTER(C.c_uint64)]),
    ("MsneSetBuildQuality", None, [_vp, C.c_int]),
    ("MsneGetTraversalCounters", C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    ("MsneGetTraversalLaneUse", C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    ("MsneTraceRays", C.c_int, [_vp, _vp, _u32, C.c_int, _vp, _vp]),
    ("MsnePick", C.c_int, [_vp, _u32, _u32, F32x2, _vp]),
    ("MsneGetEnvSize", _u32, [_vp]),
    ("MsneReadEnv", C.c_int, [_vp, _vp, _vp]),
    ("MsneGetAliasTable", _u32, [_vp, _vp, _u32]),
    ("MsneGetBounceCounters", C.c_int, [_vp, _vp, _u32]),
    ("MsneGetLaunchTimes", C.c_int, [_vp, _vp, _vp, _u32]),
    ("MsneGetPackedFilmStride", C.c_uint64, [_vp, _u32]),
    ("MsneSetMaxInflight", C.c_int, [_vp, C.c_uint64]),
    ("MsneGetMaxInflight", C.c_uint64, [_vp]),
    ("MsneGroupCreate", _vp, [_vp, _u32, _u32]),
    ("MsneGroupDestroy", None, [_vp]),
    ("MsneGroupSize", _u32, [_vp]),
    ("MsneGroupContext", _vp, [_vp, _u32]),
    ("MsneGroupLoadGlb", C.c_int, [_vp, C.c_char_p, _vp]),
    ("MsneGroupSetBackgroundExr", C.c_int, [_vp, C.c_char_p]),
    ("MsneGroupSetPipeline", C.c_int, [_vp, _vp]),
    ("MsneGroupCreateSensor", C.c_int64, [_vp, Extent2D]),
    ("MsneGroupRender", C.c_int, [_vp, _u32, _u32, _u32]),
    ("MsneGroupRenderProgressive", C.c_int, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp]),
    ("MsneGroupGetStats", C.c_int, [_vp, _vp, _vp, _vp]),
    ("MsneGroupTransport", C.c_char_p, [_vp]),
    ("MsneGroupGetLastError", C.c_char_p, [_vp]),
    ("MsneShadeProbe", C.c_int, [_vp, C.c_int, _vp, _u32, _vp]),
    ("MsneReadBvh", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
]


class MoonshineError(RuntimeError):
    pass


class MsneGlbInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("meshes", "materials", "instances", "textures", "triangles", "lens")]


def exr_load(path):
    """EXR file -> (H, W, 4) float32 (the library's codec; no GPU needed)."""
    L = load_library()
    e = Extent2D(0, 0)
    if L.MsneExrLoad(path.encode(), None, C.byref(e)) != 0:
        raise MoonshineError("MsneExrLoad: %s" % L.MsneGetIoError().decode())
    out = np.zeros((e.height, e.width, 4), np.float32)
    if L.MsneExrLoad(path.encode(), out.ctypes.data_as(C.c_void_p), C.byref(e)) != 0:
        raise MoonshineError("MsneExrLoad: %s" % L.MsneGetIoError().decode())
    return out


def exr_save(path, rgba):
    L = load_library()
    a = np.ascontiguousarray(rgba, np.float32)
    if L.MsneExrSave(path.encode(), a.ctypes.data_as(C.c_void_p), Extent2D(a.shape[1], a.shape[0])) != 0:
        raise MoonshineError("MsneExrSave: %s" % L.MsneGetIoError().decode())


def load_library(path=None):
    """dlopen the HIP library and bind every declared symbol.  Raises if the library was not built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise MoonshineError("libmoonshine_amd.so not built (%s); run `python -m moonshine_amd.build` — there is no CPU fallback" % p)
    L = C.CDLL(p)
    for name, res, args in SYMBOLS:
        if os.environ.get("MSNE_LIB") and not hasattr(L, name):
            continue             # an experimental build from an older revision (A/B timing only)
        f = getattr(L, name)   # AttributeError if the .so does not export it
        f.restype = res
        f.argtypes = args
    if path is None:
        _LIB = L
    return L


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


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
    """One HdMoonshine context = one GPU (one process per GPU in multi-GPU runs)."""

    make_lens = staticmethod(make_lens)

    def __init__(self, device=-1, tile_size=0, shard_index=0, shard_count=1):
        self.L = load_library()
        cfg = MsneConfig(device, tile_size, shard_index, shard_count)
        self.h = self.L.MsneCreate(C.byref(cfg))
        if not self.h:
            raise MoonshineError("MsneCreate failed: %s" % (self.L.MsneGetLastError(None) or b"").decode())
        self.tile_size, self.shard_index, self.shard_count = tile_size or tiles.DEFAULT_TILE, shard_index, shard_count
        self._extents = {}

    def _err(self, what):
        raise MoonshineError("%s: %s" % (what, (self.L.MsneGetLastError(self.h) or b"").decode()))

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):      # a group's members are destroyed by the group
                self.L.HdMoonshineDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene ----
    def create_mesh(self, positions, indices, normals=None, texcoords=None):
        p = _f32(positions, (-1, 3)); i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        n = _f32(normals, (-1, 3)) if normals is not None else None
        t = _f32(texcoords, (-1, 2)) if texcoords is not None else None
        ac = len(n) if n is not None else (len(t) if t is not None else 0)
        h = self.L.MsneCreateMesh(self.h, _ptr(p), _ptr(n) if n is not None else None, _ptr(t) if t is not None else None,
                                  len(p), ac, _ptr(i), len(i))
        if h < 0:
            self._err("MsneCreateMesh")
        return int(h)

    def solid_texture(self, *v):
        if len(v) == 1:
            return int(self.L.HdMoonshineCreateSolidTexture1(self.h, v[0], b""))
        if len(v) == 2:
            return int(self.L.HdMoonshineCreateSolidTexture2(self.h, F32x2(*v), b""))
        return int(self.L.HdMoonshineCreateSolidTexture3(self.h, F32x3(*v), b""))

    def create_texture(self, data, width, height, fmt):
        a = np.ascontiguousarray(data)
        h = self.L.MsneCreateTexture(self.h, _ptr(a), Extent2D(width, height), FORMATS[fmt])
        if h < 0:
            self._err("MsneCreateTexture")
        return int(h)

    def create_material(self, type, normal, emissive, color=0, metalness=0, roughness=0, ior=1.5):
        d = MsneMaterialDesc(normal, emissive, type, color, metalness, roughness, ior)
        h = self.L.MsneCreateMaterial(self.h, C.byref(d))
        if h < 0:
            self._err("MsneCreateMaterial")
        return int(h)

    def create_instance(self, geometries, transform=None, visible=True):
        arr = (Geometry * len(geometries))(*[Geometry(m, mat, bool(s)) for (m, mat, s) in geometries])
        return int(self.L.HdMoonshineCreateInstance(self.h, mat3x4(transform), arr, len(geometries), visible))

    def set_instance_transform(self, h, transform):
        self.L.HdMoonshineSetInstanceTransform(self.h, h, mat3x4(transform))

    def set_instance_visibility(self, h, v):
        self.L.HdMoonshineSetInstanceVisibility(self.h, h, v)

    def set_geometry_material(self, instance, geometry_index, material):
        """Accel.recordUpdateSingleMaterial (Accel.zig:609-628): takes effect at the next render; the caller clears the sensor."""
        if self.L.MsneSetGeometryMaterial(self.h, instance, geometry_index, material) != 0:
            self._err("MsneSetGeometryMaterial")

    def set_pipeline(self, samples_per_run=1, max_bounces=4, env_samples_per_bounce=1, mesh_samples_per_bounce=1,
                     flip_image=True, indexed_attributes=True, two_component_normal_texture=True):
        o = MsnePipelineOpts(samples_per_run, max_bounces, env_samples_per_bounce, mesh_samples_per_bounce,
                             int(flip_image), int(indexed_attributes), int(two_component_normal_texture))
        if self.L.MsneSetPipeline(self.h, C.byref(o)) != 0:
            self._err("MsneSetPipeline")

    def set_background(self, rgba, width, height):
        a = _f32(rgba, (height, width, 4))
        if self.L.MsneSetBackground(self.h, _ptr(a), Extent2D(width, height)) != 0:
            self._err("MsneSetBackground")

    def load_glb(self, path):
        """World.fromGlb + Lens.fromGlb: returns (lens handle, info dict)."""
        info = MsneGlbInfo()
        if self.L.MsneLoadGlb(self.h, path.encode(), C.byref(info)) != 0:
            raise MoonshineError("MsneLoadGlb(%s): %s" % (path, self.L.MsneGetIoError().decode()))
        return int(info.lens), {n: int(getattr(info, n)) for n, _ in MsneGlbInfo._fields_}

    def set_background_exr(self, path):
        if self.L.MsneSetBackgroundExr(self.h, path.encode()) != 0:
            raise MoonshineError("MsneSetBackgroundExr(%s): %s" % (path, self.L.MsneGetIoError().decode()))

    def save_exr(self, sensor, path):
        w, h = self._extents[sensor]
        if self.L.MsneSaveSensorExr(self.h, sensor, Extent2D(w, h), path.encode()) != 0:
            raise MoonshineError("MsneSaveSensorExr(%s): %s" % (path, self.L.MsneGetIoError().decode()))

    def create_sensor(self, width, height):
        s = int(self.L.HdMoonshineCreateSensor(self.h, Extent2D(width, height)))
        if s == 0xFFFFFFFF:
            self._err("HdMoonshineCreateSensor")
        self._extents[s] = (width, height)
        return s

    def create_lens(self, lens):
        return int(self.L.HdMoonshineCreateLens(self.h, lens))

    def set_lens(self, h, lens):
        self.L.HdMoonshineSetLens(self.h, h, lens)

    # ---- render ----
    def render(self, sensor, lens, launches=1, readback=True):
        if self.L.MsneRender(self.h, sensor, lens, launches, int(readback)) != 0:
            self._err("MsneRender")

    def reserve(self, sensor, launches):
        if self.L.MsneReserve(self.h, sensor, launches) != 0:
            self._err("MsneReserve")

    def clear_sensor(self, sensor):
        self.L.MsneClearSensor(self.h, sensor)

    def sample_count(self, sensor):
        return int(self.L.MsneGetSampleCount(self.h, sensor))

    def sensor_data(self, sensor):
        w, h = self._extents[sensor]
        p = self.L.HdMoonshineGetSensorData(self.h, sensor)
        return np.ctypeslib.as_array(p, shape=(h, w, 4)).copy()

    # ---- sharded film ----
    def packed_film(self, sensor):
        """(device pointer, float4 count) of this shard's packed film, padded to the largest shard."""
        w, h = self._extents[sensor]
        ts = self.tile_size
        total = ((w + ts - 1) // ts) * ((h + ts - 1) // ts)
        per = (total + self.shard_count - 1) // self.shard_count
        return int(self.L.MsneGetPackedFilmDevicePtr(self.h, sensor)), per * ts * ts

    def unpack_gathered(self, sensor, device_ptr, shard_count):
        if self.L.MsneUnpackGatheredFilm(self.h, sensor, C.c_void_p(device_ptr), shard_count) != 0:
            self._err("MsneUnpackGatheredFilm")

    # ---- statistics / diagnostics ----
    def accel_stats(self):
        out = (C.c_uint64 * 2)()
        self.L.MsneGetAccelStats(self.h, out)
        return {"rebuilds": int(out[0]), "tlas_updates": int(out[1])}

    def texel_pool_bytes(self):
        return int(self.L.MsneGetTexelPoolBytes(self.h))

    def set_build_quality(self, prefer_fast_trace=True):
        self.L.MsneSetBuildQuality(self.h, int(prefer_fast_trace))

    def set_profiling(self, kernel_events=True, traversal_counters=False):
        self.L.MsneSetProfiling(self.h, int(kernel_events), int(traversal_counters))

    def stats(self):
        s = MsneStats()
        if self.L.MsneGetStats(self.h, C.byref(s)) != 0:
            self._err("MsneGetStats")
        return {n: getattr(s, n) for n, _ in MsneStats._fields_}

    def reset_stats(self):
        self.L.MsneResetStats(self.h)

    def set_max_inflight(self, paths):
        """most paths traced concurrently (batches of launches are cut to fit; the default is 160 Mi)"""
        if self.L.MsneSetMaxInflight(self.h, int(paths)) != 0:
            self._err("MsneSetMaxInflight")

    def max_inflight(self):
        return int(self.L.MsneGetMaxInflight(self.h))

    def counters(self):
        s = self.stats()
        return {"closest_rays": s["closest_rays"], "shadow_rays": s["shadow_rays"], "samples": s["samples"]}

    def reset_counters(self):
        self.reset_stats()

    def traversal_counters(self):
        out = (C.c_uint64 * 20)()
        self.L.MsneGetTraversalCounters(self.h, out)
        names = ("pop", "refill", "space_changes", "node", "tri", "active_lanes", "node_lane_steps", "iterations")
        return {"closest_node_visits": int(out[0]), "closest_tri_tests": int(out[1]),
                "shadow_node_visits": int(out[2]), "shadow_tri_tests": int(out[3]),
                "closest_profile": {n: int(out[4 + i]) for i, n in enumerate(names)},
                "shadow_profile": {n: int(out[12 + i]) for i, n in enumerate(names)}}

    def traversal_lane_use(self):
        out = (C.c_uint64 * 24)()
        self.L.MsneGetTraversalLaneUse(self.h, out)
        names = ("iterations", "with_ray", "node_body", "tri_body", "space_body", "wait_space", "no_body", "iter_node", "iter_tri", "iter_space", "wait_tri_queue", "unused")
        return {"closest": {n: int(out[i]) for i, n in enumerate(names)}, "shadow": {n: int(out[12 + i]) for i, n in enumerate(names)}}

    def pick(self, sensor, lens, x, y):
        """ObjectPicker.getClickedObject: (instance, geometry, primitive, (u, v)) under normalized sensor coordinates, or None."""
        cd = MsneClickData()
        if self.L.MsnePick(self.h, sensor, lens, F32x2(x, y), C.byref(cd)) != 0:
            self._err("MsnePick")
        if cd.instance_index < 0:
            return None
        return cd.instance_index, cd.geometry_index, cd.primitive_index, (cd.barycentrics.x, cd.barycentrics.y)

    def trace_rays(self, rays, any_hit=False):
        r = _f32(rays, (-1, 7))
        ids = np.zeros((len(r), 4), np.uint32); tuv = np.zeros((len(r), 3), np.float32)
        if self.L.MsneTraceRays(self.h, _ptr(r), len(r), int(any_hit), _ptr(ids), _ptr(tuv)) != 0:
            self._err("MsneTraceRays")
        return ids, tuv

    def env(self):
        s = int(self.L.MsneGetEnvSize(self.h))
        sizes = []
        d = s
        while True:
            sizes.append(d)
            if d == 1:
                break
            d //= 2
        rgb = np.zeros((s, s, 4), np.float32); lum = np.zeros(sum(x * x for x in sizes), np.float32)
        if self.L.MsneReadEnv(self.h, _ptr(rgb), _ptr(lum)) != 0:
            self._err("MsneReadEnv")
        out, o = [], 0
        for x in sizes:
            out.append(lum[o:o + x * x].reshape(x, x).copy()); o += x * x
        return rgb, out

    def alias_table(self):
        dt = np.dtype([("alias", "<u4"), ("select", "<f4"), ("instance", "<u4"), ("geometry", "<u4"), ("primitive", "<u4")])
        n = int(self.L.MsneGetAliasTable(self.h, None, 0))
        buf = np.zeros(n, dt)
        self.L.MsneGetAliasTable(self.h, _ptr(buf), n)
        return buf

    def bounce_counters(self, max_bounces=16):
        out = np.zeros((max_bounces, 4), np.uint32)
        n = self.L.MsneGetBounceCounters(self.h, _ptr(out), max_bounces)
        if n < 0:
            self._err("MsneGetBounceCounters")
        return out[:n]

    def launch_times(self, max_launches=4096):
        """[(kind, ms)] of the last render made with kernel events on: 0 = k_trace_closest, 1 = k_trace_shadow, 2 = k_shade, in issue order (closest(b), shade(b), shadow(b), ...)"""
        kinds = np.zeros(max_launches, np.int32); ms = np.zeros(max_launches, np.float32)
        n = self.L.MsneGetLaunchTimes(self.h, _ptr(kinds), _ptr(ms), max_launches)
        return [(int(kinds[i]), float(ms[i])) for i in range(max(n, 0))]

    def shade_probe(self, fn, win, wout, x):
        """batch probe of the device shading functions (MsneShadeProbe): x (n, win) float32 -> (n, wout) float32"""
        x = _f32(x, (-1, win)); out = np.zeros((len(x), wout), np.float32)
        if self.L.MsneShadeProbe(self.h, fn, _ptr(x), len(x), _ptr(out)) != 0:
            self._err("MsneShadeProbe")
        return out

    def read_bvh(self):
        nn, nt, ni, root = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        if self.L.MsneReadBvh(self.h, None, C.byref(nn), None, C.byref(nt), C.byref(root), None, C.byref(ni)) != 0:
            self._err("MsneReadBvh")
        nodes = np.zeros(nn.value * 80, np.uint8); tris = np.zeros(nt.value * 12, np.uint32); items = np.zeros(max(ni.value, 1), np.uint32)
        if self.L.MsneReadBvh(self.h, _ptr(nodes), C.byref(nn), _ptr(tris), C.byref(nt), C.byref(root), _ptr(items), C.byref(ni)) != 0:
            self._err("MsneReadBvh")
        return nodes.reshape(-1, 80), tris.reshape(-1, 12), int(root.value), items[:ni.value]


class Group:
    """MsneGroup: one context per GPU in this process, tiles sharded over them, one gather of the films (csrc/group.hip).
    `members` are Context objects that share the group's handles (scene calls go to every one of them)."""

    def __init__(self, devices, tile_size=0):
        self.L = load_library()
        d = (C.c_int32 * len(devices))(*devices)
        self.h = self.L.MsneGroupCreate(d, len(devices), tile_size)
        if not self.h:
            raise MoonshineError("MsneGroupCreate failed: %s" % (self.L.MsneGroupGetLastError(None) or b"").decode())
        self.members = []
        for i in range(len(devices)):
            c = Context.__new__(Context)
            c.L = self.L; c.h = self.L.MsneGroupContext(self.h, i); c._extents = {}; c._borrowed = True
            c.tile_size, c.shard_index, c.shard_count = tile_size or tiles.DEFAULT_TILE, i, len(devices)
            self.members.append(c)

    def _err(self, what):
        raise MoonshineError("%s: %s" % (what, (self.L.MsneGroupGetLastError(self.h) or b"").decode()))

    def build(self, builder, **kw):
        """run a scenes.* builder on every member; returns the (sensor, lens) handles (equal on all members)"""
        out = [builder(c, **kw) for c in self.members]
        assert all(o == out[0] for o in out)
        return out[0]

    def set_pipeline(self, **kw):
        for c in self.members:
            c.set_pipeline(**kw)

    def render(self, sensor, lens, launches=1):
        if self.L.MsneGroupRender(self.h, sensor, lens, launches) != 0:
            self._err("MsneGroupRender")

    def render_progressive(self, sensor, lens, frames, max_sample_count=0, gather_every=1, present=None):
        seen = []

        def cb(user, frame, rgba, count):
            w, h = self.members[0]._extents[sensor]
            img = np.ctypeslib.as_array(rgba, shape=(h, w, 4)).copy()
            seen.append((frame, count, img))
            return int(present(frame, count, img) or 0) if present else 0
        fn = PRESENT_FN(cb)
        if self.L.MsneGroupRenderProgressive(self.h, sensor, lens, frames, max_sample_count, gather_every, fn, None) != 0:
            self._err("MsneGroupRenderProgressive")
        return seen

    def sensor_data(self, sensor):
        return self.members[0].sensor_data(sensor)

    def transport(self):
        return (self.L.MsneGroupTransport(self.h) or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            for c in self.members:
                c.h = None
            self.L.MsneGroupDestroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
