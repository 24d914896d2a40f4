"""Image-level parity between two OpenEXR files: relative per-pixel L2 ‖A − B‖₂ / ‖B‖₂ over RGB —
the north star's figure (< 1e-4 against the reference renderer) — plus max abs error and the worst pixel.
    python tools/exrdiff.py ours.exr reference.exr [--tol 1e-4]
Pure Python (tests/assets.py:exr_decode): runs on a box that has the reference renderer but not this library."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from assets import exr_decode  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a"); ap.add_argument("b"); ap.add_argument("--tol", type=float, default=1e-4)
    args = ap.parse_args()
    a, b = exr_decode(open(args.a, "rb").read()), exr_decode(open(args.b, "rb").read())
    if a.shape != b.shape:
        print("extent differs: %s vs %s" % (a.shape[:2], b.shape[:2])); return 2
    d = (a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64))
    l2 = float(np.sqrt((d * d).sum()) / max(np.sqrt((b[..., :3].astype(np.float64) ** 2).sum()), 1e-300))
    worst = np.unravel_index(np.abs(d).max(axis=2).argmax(), d.shape[:2])
    print("relative L2 %.3e  max abs %.3e at (x=%d, y=%d)  identical pixels %.2f%%" % (
        l2, float(np.abs(d).max()), worst[1], worst[0], 100.0 * float((np.abs(d).max(axis=2) == 0).mean())))
    return 0 if l2 < args.tol else 1


if __name__ == "__main__":
    sys.exit(main())
