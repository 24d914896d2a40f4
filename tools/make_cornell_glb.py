"""BASELINE.json configs[1]: writes the Cornell box as a GLB and its (black) environment as an EXR.
    python tools/make_cornell_glb.py cornell.glb black.exr  &&  moonshine_amd/offline cornell.glb black.exr out.exr 64 --width 512 --height 512 --max-bounces 8 --env-samples 0"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import io_common  # noqa: E402

if __name__ == "__main__":
    io_common.write_cornell(sys.argv[1], sys.argv[2])
