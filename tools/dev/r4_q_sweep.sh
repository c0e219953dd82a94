# development aid: sweep of the small-tile thresholds of the F(2x2) and 1x1 kernels in the training step (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 64 16; do
for Q in 512 768 1100 2100; do
for R in 512 1100; do
echo "B=$B Q_TARGET=$Q RT_TARGET=$R: $(DM_WINO_Q_TARGET_WGS=$Q DM_PW_RT_TARGET_WGS=$R python3 tools/train_time.py --batch $B --full-only 2>&1 | tail -1 | cut -c60-100)"
done
done
done
