#!/bin/bash
# Round 6, GPU session F: an overflowed plane distance decides nothing (trace.hip step_node) — the two hull-ray seeds of session E, the suite, the sweep again, what the three
# instructions cost, then the round's generation of profiles on the final kernels.
set -u
mkdir -p gpurun_out
python tools/hull_ray_explain.py 6709891 6711985 2>&1 | grep -E "checked|hip closest" | cut -c1-160 > gpurun_out/r06f_hull.txt; cat gpurun_out/r06f_hull.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r06f_suite.txt; tail -3 gpurun_out/r06f_suite.txt
bash tools/fuzz_sweep.sh 6704001 6724000 900 rays_at_the_hulls
bash tools/fuzz_sweep.sh 6801301 6802300 300 camera_far_outside
bash tools/fuzz_sweep.sh 6436001 6446000 400 random_scenes
python tools/variant_rates.py --scenes s1k20,s1,s2 default binf default binf 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06f_variant_rates.txt
bash tools/profile_generation.sh r06 2>&1 | tail -3
