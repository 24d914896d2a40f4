# usage (GPU box): bash tools/profile_generation.sh TAG — one generation of the round's committed evidence, all from the code as it is:
#   counters + kernel stats of bench.py at the driver's --steps 20 for S1 / S1-sky / S2 / the configs[2] stand-in (tools/profile_round.sh, every PMC pass its own run),
#   bench lines at 20 and 64 steps, the other scenes' rates, builder timings, a soak of six bench runs.  Scratch under gpurun_out/; tools/profile_counters.py
#   and `cp` turn it into profiles/TAG_*.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
for sc in s1 s1_sky s2 standin; do bash $R/tools/profile_round.sh $TAG $sc trace,fetch,write,sq,valu,mem,mem2 --steps 20 --warmup 4 > $O/${TAG}_profile_$sc.log 2>&1 < /dev/null; done
cd $R
timeout 300 python3 bench.py --steps 20 --warmup 4 > $O/${TAG}_bench_n1_k20.json 2> $O/${TAG}_bench_n1_k20.err < /dev/null
timeout 300 python3 bench.py --steps 64 --warmup 4 --no-other-configs > $O/${TAG}_bench_n1_k64.json 2> /dev/null < /dev/null
timeout 300 python3 bench.py --scene s2 --steps 64 --warmup 4 --no-cpu-baseline --sustain-seconds 0 > $O/${TAG}_bench_s2_k64.json 2> /dev/null < /dev/null
timeout 300 python3 bench.py --env sky --steps 64 --warmup 4 --no-cpu-baseline --sustain-seconds 0 > $O/${TAG}_bench_s1_sky_k64.json 2> /dev/null < /dev/null
timeout 300 python3 bench.py --scene s2 --steps 20 --warmup 4 --no-cpu-baseline --sustain-seconds 0 > $O/${TAG}_bench_s2_k20.json 2> /dev/null < /dev/null
timeout 300 python3 bench.py --env sky --steps 20 --warmup 4 --no-cpu-baseline --sustain-seconds 0 > $O/${TAG}_bench_s1_sky_k20.json 2> /dev/null < /dev/null
# a render is seconds of launches, the timed region a burst: the same batch back to back for 12 s (rate per 1-s window, shader clock probed while it runs)
(for a in "--steps 20" "--steps 64" "--scene s2 --steps 64"; do timeout 300 python3 bench.py $a --warmup 4 --no-cpu-baseline --no-other-configs --sustain-seconds 12 2>/dev/null < /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py $a --sustain-seconds 12: burst %.1f Mrays/s (median of %d repeats) | sustained %s' % (d['value'], d['repeats'], json.dumps(d['sustained'])))"; done) > $O/${TAG}_sustained.txt
(timeout 600 python3 tools/scene_rates.py 2>&1 < /dev/null | grep Mrays; timeout 300 python3 tools/standin_rates.py 2>&1 < /dev/null | grep "^run") > $O/${TAG}_scene_rates.txt
(for i in 1 2 3 4 5 6; do timeout 200 python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-configs --sustain-seconds 0 2>/dev/null < /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i: %.1f Mrays/s, repeats %s' % (d['value'], ' '.join('%.0f' % v for v in d['repeat_values'])))"; done) > $O/${TAG}_soak.txt
export MSNE_BUILD_TIMING=1
(for w in s1 s2 big; do timeout 300 python3 tools/build_only.py $w 2 2>&1 < /dev/null | grep -v "^$\|amdgpu.ids"; done
 timeout 300 python3 tools/many_meshes.py 3000 3 2>&1 < /dev/null | grep "meshes x\|rebuild\|swept"
 timeout 300 python3 tools/tlas_rebuild_time.py 2>&1 < /dev/null | grep "verts\|100002") > $O/${TAG}_build_timing.txt
ls -la $O/${TAG}_* | head -40
