#!/bin/bash
# Round 6, GPU session E: the long sweeps on the final tree (every line lands in profiles/r06_fuzz_sweeps.txt section 5).
set -u
mkdir -p gpurun_out
bash tools/fuzz_sweep.sh 6222001 6242000 1500 films_of_hull
bash tools/fuzz_sweep.sh 6416001 6436000 700 random_scenes
bash tools/fuzz_sweep.sh 6509001 6529000 700 random_edits
bash tools/fuzz_sweep.sh 6306001 6316000 500 lattice_rays
bash tools/fuzz_sweep.sh 6704001 6714000 500 rays_at_the_hulls
bash tools/fuzz_sweep.sh 6800301 6801300 400 camera_far_outside
bash tools/fuzz_sweep.sh 6600301 6601300 600 random_hydra
bash tools/fuzz_sweep.sh 6900000 6900600 900 random_big
bash tools/fuzz_sweep.sh 7000000 7002000 600 random_glbs tests/test_gpu_io.py
timeout 300 python3 bench.py --steps 20 --warmup 4 > gpurun_out/r06_bench_n1_k20.json 2> gpurun_out/r06_bench_n1_k20.err < /dev/null
timeout 300 python3 bench.py --steps 64 --warmup 4 --no-other-configs > gpurun_out/r06_bench_n1_k64.json 2> /dev/null < /dev/null
timeout 300 python3 bench.py > gpurun_out/r06_bench_default.json 2> /dev/null < /dev/null
python3 - <<'PY'
import json
for f in ("r06_bench_n1_k20", "r06_bench_n1_k64", "r06_bench_default"):
    d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1]); r = d["roofline"]
    print(f, round(d["value"], 1), "steps", d["steps"], "stale", r.get("counters_stale"), "bound", r["bound"], "frac %.3f" % r["frac"], "l1_tag", (r.get("l1_tag") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
