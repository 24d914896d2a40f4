#!/bin/bash
# registers / spills / occupancy of every kernel of one source, as the compiler reports them:  bash tools/kernel_resources.sh trace.hip [extra flags]
src=$1; shift
extra=""
[ "$src" = "trace.hip" ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize $extra "$@" -Rpass-analysis=kernel-resource-usage \
  -c "$(dirname "$0")/../moonshine_amd/csrc/$src" -o /dev/null 2>&1 | python3 -c '
import re, sys
name = None; row = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        if name: print(name, row)
        name = m.group(1); row = {}
    for k in ("VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPR Spill", "SGPR Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(k) + r": (\d+)", line)
        if m: row[k.split(" [")[0]] = int(m.group(1))
if name: print(name, row)
'
