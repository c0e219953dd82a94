# development aid: sweep of the weight-gradient split target at two batches (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 16 64; do
for T in 256 384 512 768 1024; do
echo "B=$B DM_WGRAD_TARGET_WGS=$T: $(DM_WGRAD_TARGET_WGS=$T python3 tools/train_time.py --batch $B --full-only 2>&1 | tail -1)"
done
done
