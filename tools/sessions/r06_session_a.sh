#!/bin/bash
# Round 6, GPU session A: the new tests first, the whole suite, the sweeps, the gather microbenchmark with its counters, the bench line as the driver runs it.
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "camera_far_outside or films_of_hull or batch_with_more_than_2_28 or bench_starts_its_own_ranks or bench_gather_path" -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r06a_new_tests.txt; tail -5 gpurun_out/r06a_new_tests.txt
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r06a_suite.txt; tail -4 gpurun_out/r06a_suite.txt
bash tools/fuzz_sweep.sh 6206001 6214000 600 films_of_hull
bash tools/fuzz_sweep.sh 6800000 6800400 600 camera_far_outside
bash tools/fuzz_sweep.sh 6410000 6413000 400 random_scenes
bash tools/run_gather_microbench.sh r06
timeout 600 python bench.py --steps 20 > gpurun_out/r06a_bench_k20.json 2> gpurun_out/r06a_bench_k20.err; cut -c1-300 gpurun_out/r06a_bench_k20.json
timeout 300 python bench.py --steps 64 --env sky --no-cpu-baseline --no-other-configs --sustain-seconds 0 > gpurun_out/r06a_bench_sky_k64.json 2>/dev/null
timeout 300 python bench.py --steps 64 --no-cpu-baseline --no-other-configs --sustain-seconds 0 > gpurun_out/r06a_bench_k64.json 2>/dev/null
python - <<'PY'
import json
for f in ("r06a_bench_k20","r06a_bench_sky_k64","r06a_bench_k64"):
    try:
        d=json.loads([l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["value"],1), d["kernel_ms_stream_order"], {k:round(v["value"],1) for k,v in d.get("other_configs",{}).items()}, (d.get("sustained") or {}).get("mrays_per_s"))
    except Exception as e: print(f, "failed", e)
PY
