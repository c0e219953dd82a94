cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_hip_train.py tests/test_hip_train_ops.py tests/test_hip_forced_dispatch.py -x -q 2>&1 | grep -v "amdgpu.ids" | tail -3
python3 tools/train_time.py --batch 64 --steps 30 --dropout 0.1 | tail -1
python3 tools/train_time.py --batch 16 --steps 30 | tail -1
