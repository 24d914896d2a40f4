"""CPU tests: the C-ABI library loads and exports every symbol include/moonshine_amd.h declares; struct layouts match
hydra/moonshine.h.  No compute call is made (there is no GPU here, and there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(gpu_api):
    L = gpu_api.load_library()
    hdr = open(os.path.join(ROOT, "include", "moonshine_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:HdMoonshine|Msne)[A-Za-z0-9]+)\s*\(", hdr))
    assert len([d for d in declared if d.startswith("HdMoonshine")]) == 24        # hydra/moonshine.h:72-95
    bound = {s[0] for s in gpu_api.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(L, name), name


# diagnostics (header part 3) a renderer front end never calls: the Zig binding leaves them out
_ZIG_LEAVES_OUT = {"MsneSetProfiling", "MsneGetAccelStats", "MsneGetTexelPoolBytes", "MsneSetBuildQuality", "MsneGetTraversalCounters", "MsneGetTraversalLaneUse",
                   "MsneGetBounceCounters", "MsneGetLaunchTimes", "MsneTraceRays", "MsneShadeProbe", "MsneGetEnvSize", "MsneReadEnv", "MsneGetAliasTable", "MsneReadBvh", "MsneProbeClockGhz"}


def _split_args(a):
    out, depth, cur = [], 0, ""
    for ch in a:
        if ch in "([{": depth += 1
        if ch in ")]}": depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip() and cur.strip() != "void":
        out.append(cur)
    return out


def test_zig_binding_declares_the_headers_entry_points():
    """moonshine_amd/host/zig/amd.zig (the file INTEGRATION.md section 2 hands a maintainer of the reference; no Zig toolchain in this image, so it is NOT compiled here):
    every entry point of parts 1 and 2 of the header is declared `pub extern fn` with the header's number of arguments, and nothing is declared that the header lacks"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "moonshine_amd.h")).read(), flags=re.S)
    c_decl = {m.group(1): len(_split_args(m.group(2))) for m in re.finditer(r"\b((?:HdMoonshine|Msne)[A-Za-z0-9]+)\s*\(([^;{]*)\)\s*;", hdr) if not m.group(1).endswith("Fn")}
    zig = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "moonshine_amd", "host", "zig", "amd.zig")).read())
    z_decl = {m.group(1): len(_split_args(m.group(2))) for m in re.finditer(r"pub extern fn (\w+)\(([^;]*)\)\s*[^;]*;", zig)}
    assert set(z_decl) <= set(c_decl), set(z_decl) - set(c_decl)
    assert set(c_decl) - set(z_decl) == _ZIG_LEAVES_OUT, (set(c_decl) - set(z_decl)) ^ _ZIG_LEAVES_OUT
    for name, n in z_decl.items():
        assert c_decl[name] == n, (name, c_decl[name], n)
    for f in ("offline.zig", "furnace_test.zig"):   # what they call exists in the binding
        src = open(os.path.join(ROOT, "moonshine_amd", "host", "zig", f)).read()
        for name in set(re.findall(r"amd\.((?:HdMoonshine|Msne)\w+)\(", src)):
            assert name in z_decl, (f, name)


def test_struct_layouts_match_reference_header(gpu_api):
    a = gpu_api
    assert C.sizeof(a.F32x2) == 8 and C.sizeof(a.F32x3) == 12 and C.sizeof(a.F32x4) == 16
    assert C.sizeof(a.Mat3x4) == 48                      # moonshine.h:33-35
    assert C.sizeof(a.Geometry) == 12                    # moonshine.h:37-41 (u32, u32, bool + pad)
    assert C.sizeof(a.Lens) == 48                        # moonshine.h:48-55 / Camera.zig:18-25
    assert C.sizeof(a.Material) == 24                    # moonshine.h:57-64
    assert C.sizeof(a.MsnePipelineOpts) == 28            # pipeline.zig:319-327: 7 x 4 B spec constants
    assert C.sizeof(a.Extent2D) == 8


def test_no_cpu_fallback_without_gpu(gpu_api):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gpu_api.MoonshineError):
        gpu_api.Context()          # must fail loudly, not fall back to a CPU path
    L = gpu_api.load_library()
    assert L.HdMoonshineCreate() in (None, 0)            # hydra.zig:107-143: NULL on init failure
    assert b"HIP device" in L.MsneGetLastError(None)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under moonshine_amd/ or include/ may import, include, link or load it"""
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#include\s*[\"<][^\">]*orc_)|(liborc)|(\boracle[./]orc)", re.M)
    for top in ("moonshine_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                    src = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert not pat.search(src), (f, pat.search(src).group(0))


REF_HEADER = "/root/reference/hydra/moonshine.h"


@pytest.mark.skipif(not os.path.exists(REF_HEADER), reason="needs the reference checkout (not present on the GPU box)")
def test_translation_unit_built_against_the_reference_header_links_and_loads(tmp_path, gpu_api):
    """drop-in at the link level: a C++ file that sees ONLY the reference's hydra/moonshine.h (the header hydra/*.cpp
    include) and calls all 24 entry points links against libmoonshine_amd.so and starts (nothing is executed: no GPU here)."""
    import subprocess
    from moonshine_amd import build as b
    src = tmp_path / "hydra_tu.cpp"
    src.write_text('''#include "%s"
int main(int argc, char** argv) {
    if (argc < 1000) return 0;            // never true: the calls below only have to compile and link
    HdMoonshine* c = HdMoonshineCreate();
    F32x3 p[3] = {}; F32x2 t[3] = {}; U32x3 i[1] = {}; uint8_t px[8] = {};
    MeshHandle m = HdMoonshineCreateMesh(c, p, p, t, 3, i, 1);
    ImageHandle i1 = HdMoonshineCreateSolidTexture1(c, 0.5f, "a"), i2 = HdMoonshineCreateSolidTexture2(c, F32x2{0, 0}, "b");
    ImageHandle i3 = HdMoonshineCreateSolidTexture3(c, F32x3{0, 0, 0}, "c"), i4 = HdMoonshineCreateRawTexture(c, px, Extent2D{1, 1}, u8x4_srgb, "d");
    MaterialHandle mat = HdMoonshineCreateMaterial(c, Material{i2, i3, i4, i1, i1, 1.5f});
    HdMoonshineSetMaterialNormal(c, mat, i2); HdMoonshineSetMaterialEmissive(c, mat, i3); HdMoonshineSetMaterialColor(c, mat, i3);
    HdMoonshineSetMaterialMetalness(c, mat, i1); HdMoonshineSetMaterialRoughness(c, mat, i1); HdMoonshineSetMaterialIOR(c, mat, 1.3f);
    Geometry g{m, mat, false};
    Mat3x4 x{};
    InstanceHandle in = HdMoonshineCreateInstance(c, x, &g, 1, true);
    HdMoonshineSetInstanceTransform(c, in, x); HdMoonshineSetInstanceVisibility(c, in, false); HdMoonshineDestroyInstance(c, in);
    SensorHandle s = HdMoonshineCreateSensor(c, Extent2D{4, 4});
    Lens l{};
    LensHandle lh = HdMoonshineCreateLens(c, l); HdMoonshineSetLens(c, lh, l);
    bool ok = HdMoonshineRebuildPipeline(c) && HdMoonshineRender(c, s, lh);
    float* d = HdMoonshineGetSensorData(c, s);
    HdMoonshineDestroy(c);
    return ok && d ? 0 : 1;
}
''' % REF_HEADER)
    exe = tmp_path / "hydra_tu"
    libdir = os.path.dirname(b.LIB)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-o", str(exe), str(src), "-L" + libdir, "-lmoonshine_amd", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")])
    assert subprocess.run([str(exe)], timeout=120).returncode == 0
