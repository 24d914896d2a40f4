"""GPU tests of the file-level boundary: MsneLoadGlb / MsneSetBackgroundExr / MsneSaveSensorExr and the `offline` CLI
(offline/main.zig:27-203) against the oracle loaded with the same files through tests/shim."""
import ctypes as C
import os
import subprocess

from moonshine_amd.hostinfo import usable_cores

import numpy as np
import pytest

from tests import io_common as io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_config0_single_triangle(tmp_path, orc, gpu_api):
    glb, exr = str(tmp_path / "tri.glb"), str(tmp_path / "white.exr")
    io.write_single_triangle(glb, exr)
    gc = gpu_api.Context(); oc = orc.Context(threads=2)
    gl, ginfo = gc.load_glb(glb); gc.set_background_exr(exr)
    ol, oinfo = io.oracle_load(orc, oc, glb, exr)
    assert ginfo == oinfo
    imgs = []
    for c, l in ((gc, gl), (oc, ol)):
        s = c.create_sensor(64, 64)
        c.set_pipeline(samples_per_run=1, max_bounces=1, env_samples_per_bounce=1, mesh_samples_per_bounce=0)
        c.render(s, l)
        imgs.append(c.sensor_data(s))
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))


def test_offline_with_a_tiled_mipmapped_environment(tmp_path):
    """the `offline` CLI (offline/main.zig:27-50) fed the SAME HDR environment as a scanline ZIP file and as a tiled, mip-mapped PIZ file written bottom-up
    (what several HDRI tools produce; tinyexr's loader — exr.zig:109-110 — takes level 0 of those): identical output files, pixel for pixel"""
    glb, sky = str(tmp_path / "gallery.glb"), str(tmp_path / "sky.exr")
    io.write_gallery(glb, sky)
    import assets
    from moonshine_amd import api
    env = api.exr_load(sky)
    scan, tiled = str(tmp_path / "scan.exr"), str(tmp_path / "tiled.exr")
    open(scan, "wb").write(assets.exr_bytes(env, "RGB", "float", "zip"))
    open(tiled, "wb").write(assets.exr_bytes(env, "RGB", "float", "piz", tiles=(32, 16), levels="mipmap", line_order=1))
    exe = os.path.join(ROOT, "moonshine_amd", "offline")
    outs = []
    for e in (scan, tiled):
        out = str(tmp_path / (os.path.basename(e) + ".out.exr"))
        r = subprocess.run([exe, glb, e, out, "4", "--width", "96", "--height", "64", "--max-bounces", "4"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(api.exr_load(out))
    assert np.isfinite(outs[0]).all() and float(outs[0][..., :3].mean()) > 0.01
    assert np.array_equal(bits(outs[0]), bits(outs[1]))


@pytest.mark.parametrize("u32", [False, True])
def test_gallery_glb_matches_oracle(tmp_path, orc, gpu_api, u32):
    glb, exr = str(tmp_path / "gallery.glb"), str(tmp_path / "sky.exr")
    io.write_gallery(glb, exr, u32=u32)
    gc = gpu_api.Context(); oc = orc.Context(threads=usable_cores())
    gl, ginfo = gc.load_glb(glb); gc.set_background_exr(exr)
    ol, oinfo = io.oracle_load(orc, oc, glb, exr)
    assert ginfo == oinfo
    imgs = []
    for c, l in ((gc, gl), (oc, ol)):
        s = c.create_sensor(160, 96)
        c.set_pipeline(samples_per_run=2, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
        c.render(s, l, launches=3)
        imgs.append(c.sensor_data(s))
    assert np.array_equal(bits(imgs[0]), bits(imgs[1]))
    out = str(tmp_path / "out.exr")
    gc.save_exr(0, out)
    back = gpu_api.exr_load(out)
    assert np.array_equal(bits(back[..., :3]), bits(imgs[0][..., :3]))


@pytest.mark.parametrize("scene", ["gallery", "room"])
def test_glb_import_rules_against_a_second_source_on_the_gpu(tmp_path, gpu_api, scene):
    """MsneLoadGlb against tests/second_source_glb.py (World.zig:44-349 + Camera.zig:26-51 restated in Python, building the scene through the C ABI's scene calls):
    two HIP contexts, one per way of getting the scene in, bit-identical films — the import semantics are no longer compared with themselves"""
    import second_source_glb
    glb, exr = str(tmp_path / "scene.glb"), str(tmp_path / "sky.exr")
    if scene == "room":
        io.write_bathroom_standin(glb, exr, spheres=16, order=3, tex=32, env=(128, 64))
    else:
        io.write_gallery(glb, exr, u32=True, interleaved=True)
    films = []
    for how in ("MsneLoadGlb", "second source"):
        c = gpu_api.Context()
        lens = c.load_glb(glb)[0] if how == "MsneLoadGlb" else second_source_glb.load(c, glb)
        c.set_background_exr(exr)
        s = c.create_sensor(320, 200)
        c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
        c.render(s, lens, launches=8)
        films.append(c.sensor_data(s))
    assert np.isfinite(films[0]).all() and float(films[0][..., :3].mean()) > 0.01
    assert np.array_equal(bits(films[0]), bits(films[1])), "%d pixels differ" % (bits(films[0]) != bits(films[1])).any(-1).sum()


def _glb_fuzz_seeds():
    from seeds import seeds
    return seeds(list(range(8)), rotating=400)


@pytest.mark.parametrize("seed", _glb_fuzz_seeds())
def test_random_glbs_against_a_second_source_on_the_gpu(tmp_path, gpu_api, orc, seed):
    """glTF files drawn from seeds (tests/io_common.py write_random_glb) through MsneLoadGlb and through tests/second_source_glb.py on two HIP contexts, and through the
    second source on the oracle: three bit-identical films.  MSNE_FUZZ_SEEDS="a-b" sweeps a range (tools/fuzz_sweep.sh a b seconds random_glbs tests/test_gpu_io.py)"""
    import second_source_glb
    glb, exr = str(tmp_path / "scene.glb"), str(tmp_path / "sky.exr")
    io.write_random_glb(glb, exr, 100000 + seed)
    films = []
    for how in ("MsneLoadGlb", "second source", "second source on the oracle"):
        c = orc.Context(threads=usable_cores()) if "oracle" in how else gpu_api.Context()
        lens = c.load_glb(glb)[0] if how == "MsneLoadGlb" else second_source_glb.load(c, glb)
        if "oracle" in how:
            assert io.shim(orc).ShimSetBackgroundExr(C.c_void_p(c.h), exr.encode()) == 0
        else:
            c.set_background_exr(exr)
        s = c.create_sensor(64, 40)
        c.set_pipeline(samples_per_run=1, max_bounces=5, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
        c.render(s, lens, launches=2)
        films.append(c.sensor_data(s))
    for k in (1, 2):
        same = (bits(films[0]) == bits(films[k])) | (np.isnan(films[0]) & np.isnan(films[k]))
        assert same.all(), "seed %d, film %d: %d values differ" % (seed, k, int((~same).sum()))


def test_offline_cli(tmp_path, orc):
    glb, exr, out = str(tmp_path / "gallery.glb"), str(tmp_path / "sky.exr"), str(tmp_path / "out.exr")
    io.write_gallery(glb, exr)
    exe = os.path.join(ROOT, "moonshine_amd", "offline")
    assert subprocess.run([exe, glb, exr, str(tmp_path / "out.png")], capture_output=True).returncode == 2     # OnlySupportsExrOutput
    r = subprocess.run([exe, glb, exr, out, "5", "--width", "128", "--height", "72", "--max-bounces", "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for line in ("seconds to load world", "seconds to create pipeline", "seconds to render", "seconds to write exr"):   # offline/main.zig:99-202
        assert line in r.stdout
    from moonshine_amd import api
    got = api.exr_load(out)
    oc = orc.Context(threads=usable_cores())
    ol, _ = io.oracle_load(orc, oc, glb, exr)
    s = oc.create_sensor(128, 72)
    oc.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    oc.render(s, ol, launches=5)
    assert np.array_equal(bits(got[..., :3]), bits(oc.sensor_data(s)[..., :3]))


def test_config1_cornell_glb_offline_64spp(tmp_path, orc):
    """BASELINE.json configs[1] end to end: Cornell-box GLB (tools/make_cornell_glb.py) -> `offline` at 512x512, 64 spp, depth 8,
    mesh-light NEE -> EXR, bit-identical to the oracle fed the same GLB"""
    glb, exr, out = str(tmp_path / "cornell.glb"), str(tmp_path / "black.exr"), str(tmp_path / "out.exr")
    io.write_cornell(glb, exr)
    exe = os.path.join(ROOT, "moonshine_amd", "offline")
    r = subprocess.run([exe, glb, exr, out, "64", "--width", "512", "--height", "512", "--max-bounces", "8", "--env-samples", "0", "--mesh-samples", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    from moonshine_amd import api
    got = api.exr_load(out)
    oc = orc.Context(threads=usable_cores())
    ol, info = io.oracle_load(orc, oc, glb, exr)
    assert info["triangles"] == 36 and info["instances"] == 8
    s = oc.create_sensor(512, 512)
    oc.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=0, mesh_samples_per_bounce=1)
    oc.render(s, ol, launches=64)
    ref = oc.sensor_data(s)
    assert np.array_equal(bits(got[..., :3]), bits(ref[..., :3]))
    a, b_ = ref[256, 40], ref[256, 471]      # the two side walls (the importer's Z-up row order mirrors the glTF scene, World.zig:339-346)
    assert 0.05 < float(ref[..., :3].mean()) < 2.0 and (a[0] - a[1]) * (b_[0] - b_[1]) < 0      # lit; one wall red, the other green


# ---- BASELINE.json configs[2] and configs[3]: "Salle de bain" at 1080p / 256 spp on one GPU, and at 4K sharded 8 ways.  The asset
# is not available ("asset substituted"): tests/io_common.py:write_bathroom_standin writes a ~1 M-triangle TEXTURED
# interior (196 PNG textures, normal maps, metallic-roughness maps, node hierarchy, glass, emissive strength + emissive texture)
# and a 2048x1024 PIZ-compressed HDR environment. ----
def _oracle_tiles(orc, glb, exr, extent, spp, tiles, film, pipe, nan_tiles=None, second_source=()):
    """`second_source`: tiles for which the oracle is NOT fed through the product's importer (tests/shim) but through tests/second_source_glb.py — the glTF rules of
    World.zig:44-349 / Camera.zig:26-51 restated in Python: at the size of configs[2] / [3] a rule the product's importer got wrong shows as a different film"""
    tx, ty = (extent[0] + 63) // 64, (extent[1] + 63) // 64
    for t in tiles:
        oc = orc.Context(threads=usable_cores(), shard_index=t, shard_count=tx * ty)
        if t in second_source:
            import ctypes as C
            import second_source_glb
            ol = second_source_glb.load(oc, glb)
            assert io.shim(orc).ShimSetBackgroundExr(C.c_void_p(oc.h), exr.encode()) == 0
        else:
            ol, _ = io.oracle_load(orc, oc, glb, exr)
        s = oc.create_sensor(*extent)
        oc.set_pipeline(**pipe)
        oc.render(s, ol, launches=spp)
        x0, y0 = (t % tx) * 64, (t // tx) * 64
        a, b = film[y0:y0 + 64, x0:x0 + 64, :3], oc.sensor_data(s)[y0:y0 + 64, x0:x0 + 64, :3]
        # bit for bit; a NaN has to be a NaN in the same pixel and channel (its sign and payload are the processor's: x86 makes 0xffc00000, gfx950 0x7fc00000)
        same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
        assert a.size and same.all(), "tile %d of %s differs from the oracle in %d values (%d NaN here, %d there)" % (t, extent, (~same).sum(), np.isnan(a).sum(), np.isnan(b).sum())
        if nan_tiles is not None and t in nan_tiles:
            assert np.isnan(a).any() and np.array_equal(np.isnan(a), np.isnan(b))


def _nan_tiles(film, limit=6):
    """64x64 tiles of the film that hold a non-finite pixel (first `limit` of them) and the number of such pixels.  squareToEqualAreaSphereInverse takes
    sqrt(1 - |z|) (mappings.hlsl:87) and a normalised direction that points straight up or down can have |z| = 1 + 1 ulp: about one sample in 1e8 is NaN, in the
    reference as here.  The tests hand these tiles to the oracle too: the comparison is bitwise, so a NaN must sit in the SAME pixel with the same bits."""
    bad = np.argwhere(~np.isfinite(film[..., :3]).all(-1))
    tx = (film.shape[1] + 63) // 64
    return sorted({int(y) // 64 * tx + int(x) // 64 for y, x in bad})[:limit], len(bad)


def test_config2_asset_substituted_textured_interior_1080p_256spp(tmp_path, orc):
    """configs[2] (asset substituted): `offline` renders the textured interior at 1920x1080, 256 spp, full MIS, max_bounces 1024
    (the CLI's defaults, offline/main.zig:106-111); three 64x64 tiles of the EXR are bit-identical to the oracle fed the same files — one of them with the
    oracle's scene built by the SECOND glTF source instead of the product's importer"""
    glb, exr, out = str(tmp_path / "bath.glb"), str(tmp_path / "sky.exr"), str(tmp_path / "out.exr")
    meta = io.write_bathroom_standin(glb, exr)
    assert meta["triangles"] > 900000 and meta["textures"] >= 64
    exe = os.path.join(ROOT, "moonshine_amd", "offline")
    r = subprocess.run([exe, glb, exr, out, "256", "--width", "1920", "--height", "1080"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout + r.stderr
    from moonshine_amd import api
    film = api.exr_load(out)
    ok = np.isfinite(film[..., :3]).all(-1)
    nan_tiles, n_nan = _nan_tiles(film)
    assert film.shape == (1080, 1920, 4) and n_nan <= 64 and 0.05 < float(film[..., :3][ok].mean()) < 5.0
    # three fixed tiles and every tile (up to six) that holds a NaN pixel: bit-identical to the oracle, NaNs included
    _oracle_tiles(orc, glb, exr, (1920, 1080), 256, sorted(set((8 * 30 + 14, 11 * 30 + 9, 14 * 30 + 22)) | set(nan_tiles)), film,
                  dict(samples_per_run=1, max_bounces=1024, env_samples_per_bounce=1, mesh_samples_per_bounce=1), nan_tiles, second_source=(11 * 30 + 9,))
    print("configs[2]: ASSET SUBSTITUTED (Salle de bain is not available: tests/io_common.py:write_bathroom_standin); %d NaN pixels, their tiles %s equal the oracle's\n" % (n_nan, nan_tiles) + r.stdout)


def test_config3_asset_substituted_4k_1024spp_sharded_eight_members_on_one_gpu(tmp_path, orc):
    """configs[3] at its stated 3840x2160 and 1024 spp (asset substituted, and 8 members on this box's ONE GPU instead of 8 GPUs — the gather is a device copy here,
    ncclGather on distinct GPUs): `offline --devices 0,0,0,0,0,0,0,0`; three tiles of the assembled EXR, and the tiles that hold NaN pixels, are bit-identical to the oracle"""
    glb, exr, out = str(tmp_path / "bath.glb"), str(tmp_path / "sky.exr"), str(tmp_path / "out.exr")
    io.write_bathroom_standin(glb, exr)
    exe = os.path.join(ROOT, "moonshine_amd", "offline")
    r = subprocess.run([exe, glb, exr, out, "1024", "--width", "3840", "--height", "2160", "--max-bounces", "8", "--devices", "0,0,0,0,0,0,0,0"],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "on 8 GPUs (film gather: copy, 1 x" in r.stdout, r.stdout + r.stderr
    from moonshine_amd import api
    film = api.exr_load(out)
    nan_tiles, n_nan = _nan_tiles(film, limit=3)
    assert film.shape == (2160, 3840, 4) and n_nan <= 256
    _oracle_tiles(orc, glb, exr, (3840, 2160), 1024, sorted(set((17 * 60 + 28, 25 * 60 + 41, 33 * 60 + 59)) | set(nan_tiles)), film,
                  dict(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1), nan_tiles, second_source=(25 * 60 + 41,))
    print("configs[3]: ASSET SUBSTITUTED, and 8 members on ONE GPU (device-copy gather instead of ncclGather); 1024 spp; %d NaN pixels, their tiles %s equal the oracle's\n" % (n_nan, nan_tiles) + r.stdout)


def test_offline_cli_progressive_and_sharded(tmp_path, orc):
    """`offline --progressive` (the online frame loop, headless) and `--devices` (members sharing this box's GPU): frames are reported as a
    viewer would see them, sampling stops at --max-sample-count, and the EXR equals a plain 6-spp render of the oracle"""
    glb, exr, out = str(tmp_path / "gallery.glb"), str(tmp_path / "sky.exr"), str(tmp_path / "out.exr")
    io.write_gallery(glb, exr)
    exe = os.path.join(ROOT, "moonshine_amd", "offline")
    r = subprocess.run([exe, glb, exr, out, "--width", "96", "--height", "64", "--max-bounces", "8", "--devices", "0,0,0", "--progressive", "9", "--max-sample-count", "6",
                        "--present-every", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    frames = [l for l in r.stdout.splitlines() if l.startswith("frame ")]
    assert [l.split(":")[0] for l in frames] == ["frame 1", "frame 3", "frame 5", "frame 7", "frame 8"]          # every second frame and the last
    assert [int(l.split(":")[1].split()[0]) for l in frames] == [2, 4, 6, 6, 6]                                  # stops launching at max_sample_count
    assert "on 3 GPUs (film gather: copy" in r.stdout
    from moonshine_amd import api
    got = api.exr_load(out)
    oc = orc.Context(threads=usable_cores())
    ol, _ = io.oracle_load(orc, oc, glb, exr)
    s = oc.create_sensor(96, 64)
    oc.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    oc.render(s, ol, launches=6)
    assert np.array_equal(bits(got[..., :3]), bits(oc.sensor_data(s)[..., :3]))
    bad = subprocess.run([exe, glb, exr, out, "--gpus", "0"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "--gpus" in bad.stderr
    bad = subprocess.run([exe, glb, exr, out, "--devices", "0,99"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "does not exist" in bad.stderr
