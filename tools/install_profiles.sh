#!/bin/bash
# Copy the artefacts of `bash tools/final_profiles.sh <tag>` (merged back under gpurun_out/) into profiles/:
#   bash tools/install_profiles.sh <tag>
set -e
tag=$1
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
cp gpurun_out/${tag}_stats/s_kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp gpurun_out/${tag}_pmc_hbm_traffic.json profiles/${tag}_pmc_hbm_traffic.json
(cat gpurun_out/${tag}_other_configs.txt; echo; cat gpurun_out/${tag}_vae.txt) | grep -v amdgpu.ids > profiles/${tag}_other_configs.txt
cp gpurun_out/${tag}_b8/s_kernel_stats.csv profiles/${tag}_b8_64x64_kernel_stats.csv
cp gpurun_out/${tag}_train/s_kernel_stats.csv profiles/${tag}_train_kernel_stats.csv
grep -v amdgpu.ids gpurun_out/${tag}_train_time.txt > profiles/${tag}_train_time.txt
cp gpurun_out/${tag}_train_pmc.json profiles/${tag}_train_pmc.json
ls -la profiles | grep ${tag}
