# round-4 random sweeps on the GPU box: default dispatch, then the small-tile kernel forms forced everywhere
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4_fuzz_sweeps.txt
echo "# off-line random sweeps (round 4): tools/fuzz_unet.py --train (seeds 7-8 x 16 configurations), tools/fuzz_conv.py (seeds 6-7 x 250 shapes), tools/fuzz_block.py (seed 3 x 150); then the same conv / block / unet sweeps with DM_PW_RT_TARGET_WGS=1000000 DM_WINO_Q_TARGET_WGS=1000000 (small-tile kernel forms everywhere)" > $out
for s in ${UNET_SEEDS:-7 8}; do python3 tools/fuzz_unet.py --train --seed $s 2>&1 | grep -v amdgpu.ids | tail -3 >> $out; echo "unet seed $s done"; done
for s in 6 7; do python3 tools/fuzz_conv.py --seed $s --n 250 2>&1 | grep -v amdgpu.ids | tail -2 >> $out; echo "conv seed $s done"; done
python3 tools/fuzz_block.py --seed 3 --n 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
export DM_PW_RT_TARGET_WGS=1000000 DM_WINO_Q_TARGET_WGS=1000000
echo "## small-tile forms forced" >> $out
python3 tools/fuzz_conv.py --seed 8 --n 250 2>&1 | grep -v amdgpu.ids | tail -2 >> $out; echo "forced conv done"
python3 tools/fuzz_block.py --seed 4 --n 150 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
python3 tools/fuzz_unet.py --train --seed 9 2>&1 | grep -v amdgpu.ids | tail -3 >> $out
cat $out
