"""Writes the stand-in for BASELINE.json configs[2]/[3] (Salle-de-bain is not available here): a ~1 M-triangle TEXTURED interior
as a GLB + a 2048x1024 PIZ-compressed HDR environment (tests/io_common.py:write_bathroom_standin).
    python tools/make_bathroom_standin.py out.glb out.exr
    moonshine_amd/offline out.glb out.exr render.exr 256 --width 1920 --height 1080 [--gpus 8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import io_common as io
print(io.write_bathroom_standin(sys.argv[1], sys.argv[2]))
