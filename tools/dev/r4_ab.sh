# development aid: A/B of one environment switch on the training step:  bash tools/dev/r4_ab.sh VAR   (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for B in 64 16; do
echo "B=$B default: $(python3 tools/train_time.py --batch $B --full-only 2>&1 | tail -1 | cut -c60-100)"
echo "B=$B $1=1:    $(env $1=1 python3 tools/train_time.py --batch $B --full-only 2>&1 | tail -1 | cut -c60-100)"
done
done
