#!/bin/bash
# Round 6, GPU session B: the product's boxes grown by 1e-4 of their extent (every BVH changes), the oracle's box test with the same term: suite, truth table of the
# seeds the CPU sweep found, sweeps, the microbenchmark's counters again (table fixed), A/B of the environment top in LDS and of the growth.
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "camera_far_outside or films_of_hull" -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r06b_new_tests.txt; tail -3 gpurun_out/r06b_new_tests.txt
python tools/film_truth.py hull 6204351 6226272 6240180 6200851 6201195 6201640 2>&1 | tail -8 > gpurun_out/r06b_film_truth.txt; cat gpurun_out/r06b_film_truth.txt
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r06b_suite.txt; tail -4 gpurun_out/r06b_suite.txt
bash tools/fuzz_sweep.sh 6214001 6222000 600 films_of_hull
bash tools/fuzz_sweep.sh 6800000 6800300 400 camera_far_outside
bash tools/fuzz_sweep.sh 6506001 6509000 400 random_edits
bash tools/fuzz_sweep.sh 6413001 6416000 300 random_scenes
bash tools/fuzz_sweep.sh 6304001 6306000 300 lattice_rays
bash tools/fuzz_sweep.sh 6702001 6704000 300 rays_at_the_hulls
python - <<'PY'
from moonshine_amd import build as b
b.build(variant="noenvtop", extra_flags=["-DMSNE_ENV_TOP_LDS=0"]); b.build(variant="nogrow", extra_flags=["-DMSNE_BOX_GROWTH=0.0f"])
PY
python tools/variant_rates.py --scenes sky,s1k20,s1,s2 default noenvtop nogrow default noenvtop nogrow 2>&1 | tee gpurun_out/r06b_variant_rates.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/gather_microbench tools/gather_microbench.hip
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r06_gather_pmc -o p -- /tmp/gather_microbench pmc > $R/gpurun_out/r06_gather_pmc.log 2>&1 || echo "pmc pass failed"
cd $R; python3 tools/gather_counters.py gpurun_out/r06_gather_pmc gpurun_out/r06_l1_tag_calibration.json > gpurun_out/r06_gather_counters.txt 2>&1; rm -rf gpurun_out/r06_gather_pmc; tail -3 gpurun_out/r06_gather_counters.txt | cut -c1-300
