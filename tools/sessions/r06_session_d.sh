#!/bin/bash
# Round 6, GPU session D: how many node visits a popped-entry bound could save (verdict r5 item 7), measured.
set -u
mkdir -p gpurun_out
python - <<'PY'
from moonshine_amd import build as b
for k in (1, 2, 3):
    b.build(variant="cnt%d" % k, extra_flags=["-DTRACE_COUNT_EMPTY=%d" % k])
PY
(for k in 1 2 3; do MSNE_LIB=moonshine_amd/libmoonshine_amd_cnt$k.so python tools/empty_visits.py 2>/dev/null | sed "s/^/cnt$k /"; done) | tee gpurun_out/r06_stale_visits.txt
