#!/bin/bash
# Round 6, GPU session G: the last long sweeps, on the final tree.
set -u
mkdir -p gpurun_out
bash tools/fuzz_sweep.sh 6724001 6754000 1200 rays_at_the_hulls
bash tools/fuzz_sweep.sh 6242001 6262000 1500 films_of_hull
bash tools/fuzz_sweep.sh 6316001 6336000 1000 lattice_rays
bash tools/fuzz_sweep.sh 6529001 6549000 700 random_edits
bash tools/fuzz_sweep.sh 6446001 6466000 700 random_scenes
bash tools/fuzz_sweep.sh 6802301 6804300 400 camera_far_outside
bash tools/fuzz_sweep.sh 6900601 6901200 900 random_big
bash tools/fuzz_sweep.sh 6601301 6602300 600 random_hydra
