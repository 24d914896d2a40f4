#!/bin/bash
# A wider sweep of tests/test_gpu_parity.py::test_random_scenes_match_oracle (or $4 = random_edits: ::test_random_edits_match_oracle) than the suite's sixteen seeds:
# every seed in [$1, $2], all failures listed; $3 = time limit in seconds; $5 = the test file ($4 = random_glbs lives in tests/test_gpu_io.py).
#   gpurun --timeout 1500 -- 'bash tools/fuzz_sweep.sh 16 400'   -> gpurun_out/fuzz_${4:-random_scenes}_$1_$2.txt
set -u
mkdir -p gpurun_out
MSNE_FUZZ_SEEDS="$1-$2" timeout "${3:-1200}" python -m pytest "${5:-tests/test_gpu_parity.py}" -q -m gpu -k "${4:-random_scenes}" -p no:cacheprovider </dev/null 2>&1 | grep -v "^$" | tail -60 > "gpurun_out/fuzz_${4:-random_scenes}_$1_$2.txt"
tail -5 "gpurun_out/fuzz_${4:-random_scenes}_$1_$2.txt"
