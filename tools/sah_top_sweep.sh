# usage (GPU box): bash tools/sah_top_sweep.sh "4096 16384 65536 4000000" [s1|s2|sky]
# Mrays/s (bench.py, 64 steps) and BLAS build time against the number of clusters PLOC leaves for the top-down sweep ($MSNE_SAH_TOP; larger than the
# primitive count = no PLOC at all: one sweep over every primitive)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for sc in ${2:-s1 s2}; do
  for top in $1; do
    extra="--scene $sc"; [ "$sc" = sky ] && extra="--scene s1 --env sky"
    MSNE_SAH_TOP=$top MSNE_BUILD_TIMING=1 timeout 300 python3 $R/bench.py $extra --steps 64 --warmup 4 --no-cpu-baseline --repeats 3 > /tmp/o.json 2> /tmp/o.err < /dev/null
    v=$(python3 -c "import json;print('%.0f' % json.load(open('/tmp/o.json'))['value'])" 2>/dev/null)
    b=$(grep "rebuild" /tmp/o.err | head -1 | sed 's/.*BLAS \([0-9.]*\) ms.*TLAS \([0-9.]*\) ms.*/BLAS \1 ms TLAS \2 ms/')
    echo "$sc MSNE_SAH_TOP=$top: $v Mrays/s; first build: $b"
  done
done
