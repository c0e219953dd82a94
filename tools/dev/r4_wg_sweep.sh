# development aid: sweep of the grouped weight-gradient launch size at two batches (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 16 64; do
for T in 768 1024 1536 2048 3072 4096; do
echo "B=$B DM_WGRAD_GROUP_WGS=$T: $(DM_WGRAD_GROUP_WGS=$T python3 tools/train_time.py --batch $B --full-only 2>&1 | tail -1)"
done
done
