#!/bin/bash
# End-of-round measurements in one GPU call:  bash tools/final_profiles.sh <tag>   (writes gpurun_out/<tag>_*)
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== pmc"; bash tools/pmc_run.sh ${tag}_pmc
python3 tools/pmc_summary.py gpurun_out/${tag}_pmc > gpurun_out/${tag}_pmc_hbm_traffic.json
mkdir -p profiles && cp gpurun_out/${tag}_pmc_hbm_traffic.json profiles/${tag}_pmc_hbm_traffic.json
echo "== rocprof stats (ddpm1000, 1 call)"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs --no-train > gpurun_out/${tag}_stats.log 2>&1
rm -f gpurun_out/${tag}_stats/*kernel_trace.csv; ls gpurun_out/${tag}_stats
cp gpurun_out/${tag}_stats/s_kernel_stats.csv profiles/${tag}_kernel_stats.csv   # bench.py reads the family shares from it
echo "== configs"; python3 tools/config_bench.py > gpurun_out/${tag}_other_configs.txt 2>&1; tail -12 gpurun_out/${tag}_other_configs.txt
echo "== b8 stats"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_b8 -o s -- python3 tools/step_time.py --batch 8 --size 64 > gpurun_out/${tag}_b8.log 2>&1
rm -f gpurun_out/${tag}_b8/*kernel_trace.csv; tail -1 gpurun_out/${tag}_b8.log
echo "== vae"; python3 tools/vae_time.py > gpurun_out/${tag}_vae.txt 2>&1; tail -3 gpurun_out/${tag}_vae.txt
echo "== training step stats (the bench.py train leg: B=64, dropout 0.1; 1 + 9 full iterations)"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_train -o s -- python3 tools/train_time.py --batch 64 --dropout 0.1 --full-only --warmup 1 --steps 9 > gpurun_out/${tag}_train.log 2>&1
rm -f gpurun_out/${tag}_train/*kernel_trace.csv; cp gpurun_out/${tag}_train/s_kernel_stats.csv profiles/${tag}_train_kernel_stats.csv
python3 tools/train_time.py --batch 64 --dropout 0.1 --steps 20 > gpurun_out/${tag}_train_time.txt 2>&1; tail -2 gpurun_out/${tag}_train_time.txt
echo "== training step pmc (1 + 3 full iterations)"
export PMC_ITERS=4 PMC_CMD="python3 tools/train_time.py --batch 64 --dropout 0.1 --full-only --warmup 1 --steps 3"
bash tools/pmc_run.sh ${tag}_trainpmc
python3 tools/pmc_summary.py gpurun_out/${tag}_trainpmc > gpurun_out/${tag}_train_pmc.json
cp gpurun_out/${tag}_train_pmc.json profiles/${tag}_train_pmc.json
unset PMC_ITERS PMC_CMD
echo "== bench"; python3 bench.py --steps 3 --warmup 1 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 600 gpurun_out/${tag}_bench.json
