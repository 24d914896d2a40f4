#!/bin/bash
# Round 6, GPU session C: the suite with its slowest tests named, the far-camera and hull-ray sweeps after the tests' own fixes.
set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=30 2>&1 | tail -60 > gpurun_out/r06c_suite.txt; tail -45 gpurun_out/r06c_suite.txt | cut -c1-200
bash tools/fuzz_sweep.sh 6800000 6800300 400 camera_far_outside
bash tools/fuzz_sweep.sh 6702001 6704000 300 rays_at_the_hulls
