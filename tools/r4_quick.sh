set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_hip_train.py tests/test_hip_train_ops.py -x -q 2>&1 | tail -5
python3 tools/train_time.py --batch 64 --steps 20
python3 tools/train_time.py --batch 64 --steps 20 --dropout 0.1
python3 tools/train_time.py --batch 16 --steps 20
DM_TRAIN_NO_LANDING_FUSE=1 python3 tools/train_time.py --batch 64 --steps 20
