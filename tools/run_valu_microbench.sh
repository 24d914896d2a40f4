# usage (on the GPU box): bash tools/run_valu_microbench.sh   → gpurun_out/r03_valu_microbench.txt, r03_gather_microbench.txt (+ the SQ counters of the VALU binary)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/valu_microbench "$R/tools/valu_microbench.hip" || exit 1
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/gather_microbench "$R/tools/gather_microbench.hip" || exit 1
/tmp/valu_microbench > "$R/gpurun_out/r03_valu_microbench.txt" 2>&1
/tmp/gather_microbench > "$R/gpurun_out/r03_gather_microbench.txt" 2>&1
if [ "$1" = "pmc" ]; then
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/r03_valu_pmc" -o p -- /tmp/valu_microbench > "$R/gpurun_out/r03_valu_pmc.log" 2>&1 || echo pmc failed
fi
grep -v "^#" "$R/gpurun_out/r03_valu_microbench.txt" | awk '$0 ~ / 6  / || $0 ~ / 2  /' | cut -c1-130
