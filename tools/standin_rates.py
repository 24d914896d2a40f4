"""Mrays/s of the configs[2] stand-in (tests/io_common.py:write_bathroom_standin, ~1 M textured triangles in 54 transformed instances of
distinct meshes) at 1920x1080, inputs resident, warm: python tools/standin_rates.py [launches] [texture size, default 64]
(texture size 1024 / 2048: the texel pool of a real asset — 112 PNG textures of that edge length)"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch  # noqa
from tests import io_common as io
from moonshine_amd import api
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
TEX = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = tempfile.mkdtemp()
glb, exr = os.path.join(d, "bath.glb"), os.path.join(d, "sky.exr")
io.write_bathroom_standin(glb, exr, tex=TEX)
for rep in range(2):
    c = api.Context()
    lens, info = c.load_glb(glb); c.set_background_exr(exr)
    s = c.create_sensor(1920, 1080)
    c.set_pipeline(samples_per_run=1, max_bounces=8, env_samples_per_bounce=1, mesh_samples_per_bounce=1)
    c.reserve(s, K); c.set_profiling(True, False)
    c.render(s, lens, launches=4, readback=False)
    c.reset_stats(); c.render(s, lens, launches=K, readback=False)
    st = c.stats()
    rays = st["closest_rays"] + st["shadow_rays"]
    pool = c.texel_pool_bytes() if hasattr(c, "texel_pool_bytes") and hasattr(c.L, "MsneGetTexelPoolBytes") else -1
    print("run %d: tex %d texel pool %.1f MB  %.1f Mrays/s (%.2f ms; closest %.2f shadow %.2f shade %.2f ms; %d rays)" % (rep, TEX, pool / 1e6, rays / st["render_ms"] / 1e3, st["render_ms"], st["trace_closest_ms"], st["trace_shadow_ms"], st["shade_ms"], rays))
    c.close()
