# usage (on the GPU box): bash tools/run_variants.sh "SCENES" VARIANT...   — A/B rates of library variants + a quick parity check of each
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
SC=$1; shift
python3 tools/variant_rates.py --scenes "$SC" "$@" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/variants.txt
for v in "$@"; do
  [ "$v" = default ] && continue
  MSNE_LIB=$R/moonshine_amd/libmoonshine_amd_$v.so timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trace_rays or s1_small or instanced_s2_small or coincident or s1_full_frame" 2>&1 | tail -2 | sed "s/^/[$v] /" | tee -a gpurun_out/variants.txt
done
