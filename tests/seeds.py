"""Which seeds the randomized tests run.

Every randomized test has a FIXED list (seeds that once failed, and a few that always run: regressions stay covered) and, on the GPU, a ROTATING range: `count` consecutive
seeds starting at 1000 + rotation(), where rotation() is derived from the contents of the product's and the oracle's sources.  Whenever a kernel, the host code or the
oracle changes, the driver-run suite therefore covers seeds no earlier tree was tested on (the GPU box receives the tree without .git, so the tree's own bytes stand in
for `git rev-parse HEAD`); a failing test names its seed in its id, and MSNE_FUZZ_SEEDS="a-b" replays or sweeps any range (tools/fuzz_sweep.sh)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ROT = None


def rotation():
    global _ROT
    if _ROT is None:
        h = hashlib.sha256()
        for d, exts in (("moonshine_amd/csrc", (".hip", ".h")), ("moonshine_amd/host", (".cpp", ".h")), ("oracle", (".c", ".h")), ("include", (".h",))):
            for f in sorted(os.listdir(os.path.join(ROOT, d))):
                if f.endswith(exts):
                    h.update(f.encode()); h.update(open(os.path.join(ROOT, d, f), "rb").read())
        _ROT = int(h.hexdigest()[:8], 16) % 900000
    return _ROT


def seeds(fixed, rotating=0):
    """the suite's seeds: `fixed` + `rotating` consecutive seeds from 1000 + rotation(); or every seed of MSNE_FUZZ_SEEDS="a-b"."""
    spec = os.environ.get("MSNE_FUZZ_SEEDS")
    if spec:
        a, _, b = spec.partition("-")
        return list(range(int(a), int(b or a) + 1))
    base = 1000 + rotation()
    out = list(fixed)
    return out + [s for s in range(base, base + rotating) if s not in out]
