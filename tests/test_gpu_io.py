"""GPU tests of the file-level boundary: MsneLoadGlb / MsneSetBackgroundExr / MsneSaveSensorExr and the `offline` CLI
(offline/main.zig:27-203) against the oracle loaded with the same files through tests/shim."""
import os
import subprocess

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


@pytest.mark.parametrize("u32", [False, True])
def test_gallery_glb_matches_oracle(tmp_path, orc, gpu_api, u32):
    glb, exr = str(tmp_path / "gallery.glb"), str(tmp_path / "sky.exr")
    io.write_gallery(glb, exr, u32=u32)
    gc = gpu_api.Context(); oc = orc.Context(threads=os.cpu_count())
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
    oc = orc.Context(threads=os.cpu_count())
    ol, _ = io.oracle_load(orc, oc, glb, exr)
    s = oc.create_sensor(128, 72)
    oc.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    oc.render(s, ol, launches=5)
    assert np.array_equal(bits(got[..., :3]), bits(oc.sensor_data(s)[..., :3]))


@pytest.mark.skipif((os.cpu_count() or 1) < 64, reason="the oracle needs a many-core host for 16.7 M samples (the GPU box has 256 threads)")
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
    oc = orc.Context(threads=os.cpu_count())
    ol, info = io.oracle_load(orc, oc, glb, exr)
    assert info["triangles"] == 36 and info["instances"] == 8
    s = oc.create_sensor(512, 512)
    oc.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=0, mesh_samples_per_bounce=1)
    oc.render(s, ol, launches=64)
    ref = oc.sensor_data(s)
    assert np.array_equal(bits(got[..., :3]), bits(ref[..., :3]))
    a, b_ = ref[256, 40], ref[256, 471]      # the two side walls (the importer's Z-up row order mirrors the glTF scene, World.zig:339-346)
    assert 0.05 < float(ref[..., :3].mean()) < 2.0 and (a[0] - a[1]) * (b_[0] - b_[1]) < 0      # lit; one wall red, the other green
