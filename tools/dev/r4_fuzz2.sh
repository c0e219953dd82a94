# round-4 (second half) random sweeps on the GPU box: U-Net configurations incl. per-stage head counts, forward + every gradient;
# then the same with the round's new kernels switched back to their previous forms (the sweep of the other path)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4_fuzz_sweeps2.txt
echo "# off-line random sweeps (round 4, second half): tools/fuzz_unet.py --train (seeds 11-14 x 16), --big --train (seed 15 x 8); tools/fuzz_conv.py seed 9 x 250; then fuzz_unet seed 16 with the previous kernel forms (DM_LINATTN_BWD_VALU DM_LINATTN_VALU DM_TRAIN_NO_FINAL_FUSE DM_WGRAD_INIT_VALU DM_ATTN_BWD_NO_PAIRS DM_REPACK_ROT_TMP DM_LINATTN_NO_KSTATS)" > $out
for s in 11 12 13 14; do python3 tools/fuzz_unet.py --train --seed $s --n 16 2>&1 | grep -v amdgpu.ids | grep -v "^OK" | tail -4 >> $out; echo "unet seed $s done"; done
python3 tools/fuzz_unet.py --train --big --seed 15 --n 8 2>&1 | grep -v amdgpu.ids | grep -v "^OK" | tail -4 >> $out; echo "big done"
python3 tools/fuzz_conv.py --seed 9 --n 250 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
echo "## previous kernel forms" >> $out
DM_LINATTN_BWD_VALU=1 DM_LINATTN_VALU=1 DM_TRAIN_NO_FINAL_FUSE=1 DM_WGRAD_INIT_VALU=1 DM_ATTN_BWD_NO_PAIRS=1 DM_REPACK_ROT_TMP=1 DM_LINATTN_NO_KSTATS=1 python3 tools/fuzz_unet.py --train --seed 16 --n 16 2>&1 | grep -v amdgpu.ids | grep -v "^OK" | tail -4 >> $out
cat $out
