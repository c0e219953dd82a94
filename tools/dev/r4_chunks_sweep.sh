# development aid: the K-split floor of the F(2x2) / 1x1 kernels (chunks per split) in training and in small-batch sampling
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for M in 8 4 2; do
for P in 8 4; do
export DM_WINO_MIN_CHUNKS=$M DM_PW_MIN_CHUNKS=$P
echo "WINO_MIN_CHUNKS=$M PW_MIN_CHUNKS=$P"
python3 tools/train_time.py --batch 16 --full-only 2>&1 | tail -1 | cut -c60-110
python3 tools/train_time.py --batch 64 --full-only 2>&1 | tail -1 | cut -c60-110
python3 tools/step_time.py --batch 8 --size 64 2>&1 | tail -1
python3 tools/step_time.py --batch 32 --size 64 2>&1 | tail -1
done
done
