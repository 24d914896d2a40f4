#!/bin/bash
# Round 6, step 1: the flat-node margin in the tree.  The five seeds that were red, the new 2^28 test, then the two sweeps the verdict names, then the whole -m gpu suite,
# then the rates.   gpurun --timeout 2700 -- 'bash tools/r06_verify.sh'
set -u
mkdir -p gpurun_out
{
MSNE_FUZZ_SEEDS="6200053-6200053" python -m pytest tests/test_gpu_parity.py -q -m gpu -k "films_of_hull" -p no:cacheprovider 2>&1 | tail -3
for s in 6200851 6201195 6201640; do MSNE_FUZZ_SEEDS="$s-$s" python -m pytest tests/test_gpu_parity.py -q -m gpu -k "films_of_hull" -p no:cacheprovider 2>&1 | tail -1; done
MSNE_FUZZ_SEEDS="6502872-6502872" python -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_edits" -p no:cacheprovider 2>&1 | tail -1
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_batch_with_more_than_2_28 or test_sub_queues" -p no:cacheprovider 2>&1 | tail -3
} > gpurun_out/r06_red_seeds.txt 2>&1
cat gpurun_out/r06_red_seeds.txt
bash tools/fuzz_sweep.sh 6200000 6206000 900 films_of_hull
bash tools/fuzz_sweep.sh 6500000 6506000 900 random_edits
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r06_suite.txt; cat gpurun_out/r06_suite.txt
python bench.py --steps 20 > gpurun_out/r06_bench_k20_a.json 2> gpurun_out/r06_bench_k20_a.err; cut -c1-400 gpurun_out/r06_bench_k20_a.json
