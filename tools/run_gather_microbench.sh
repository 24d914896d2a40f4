#!/bin/bash
# tools/gather_microbench.hip on the GPU box: the plain table, then one counters pass (its own run: gpurun refuses --pmc next to other trace domains).
#   gpurun --timeout 900 -- 'bash tools/run_gather_microbench.sh r06'   ->  gpurun_out/TAG_gather_microbench.txt, gpurun_out/TAG_gather_counters.txt
set -u
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$R/gpurun_out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/gather_microbench "$R/tools/gather_microbench.hip" || exit 1
timeout 600 /tmp/gather_microbench > "$R/gpurun_out/${TAG}_gather_microbench.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/${TAG}_gather_pmc"
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/${TAG}_gather_pmc" -o p -- /tmp/gather_microbench pmc > "$R/gpurun_out/${TAG}_gather_pmc.log" 2>&1 || echo "pmc pass failed"
python3 "$R/tools/gather_counters.py" "$R/gpurun_out/${TAG}_gather_pmc" "$R/gpurun_out/${TAG}_l1_tag_calibration.json" > "$R/gpurun_out/${TAG}_gather_counters.txt" 2>&1
rm -rf "$R/gpurun_out/${TAG}_gather_pmc"
tail -n 40 "$R/gpurun_out/${TAG}_gather_counters.txt"
