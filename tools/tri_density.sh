# usage (GPU box): bash tools/tri_density.sh VARIANT...   — lane use of the traversal loop (STATS build, 1920x1080 x 4 launches) and rates (64 launches) of library variants, S1 and S2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"; cd "$R"
for v in "$@"; do
  lib=""; [ "$v" != default ] && lib="$R/moonshine_amd/libmoonshine_amd_$v.so"
  for sc in s1 s2; do
    echo "== $v $sc"
    MSNE_LIB=$lib SCENE=$sc python3 tools/lane_use.py 2>&1 | grep -v amdgpu.ids
  done
done
python3 tools/variant_rates.py --scenes s1,s2 "$@" 2>&1 | grep -v amdgpu.ids
