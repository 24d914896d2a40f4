#!/bin/bash
# Round 6, GPU session H: the bench lines with the committed counters of the final kernels (counters_stale false), the scaling prediction on this round's kernels, the suite once
# more (the rendezvous ports are picked free now).
set -u
mkdir -p gpurun_out
timeout 300 python3 bench.py --steps 20 --warmup 4 > gpurun_out/r06_bench_n1_k20.json 2> gpurun_out/r06_bench_n1_k20.err < /dev/null
timeout 300 python3 bench.py --steps 64 --warmup 4 --no-other-configs > gpurun_out/r06_bench_n1_k64.json 2> /dev/null < /dev/null
timeout 300 python3 bench.py > gpurun_out/r06_bench_default.json 2> /dev/null < /dev/null
(timeout 600 python3 tools/shard_time.py --steps 20 2>/dev/null | grep "^steps"; timeout 600 python3 tools/shard_time.py --steps 64 2>/dev/null | grep "^steps") > gpurun_out/r06_shard_time.txt; cat gpurun_out/r06_shard_time.txt
python3 - <<'PY'
import json
for f in ("r06_bench_n1_k20", "r06_bench_n1_k64", "r06_bench_default"):
    d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1]); r = d["roofline"]
    print(f, round(d["value"], 1), "steps", d["steps"], "stale", r.get("counters_stale"), "bound", r["bound"], "frac %.3f" % r["frac"], "l1_tag", (r.get("l1_tag") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/r06h_suite.txt; tail -3 gpurun_out/r06h_suite.txt
