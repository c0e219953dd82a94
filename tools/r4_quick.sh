set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_hip_ops.py tests/test_hip_fuzz.py tests/test_hip_configs.py tests/test_hip_train_ops.py -x -q 2>&1 | grep -v "amdgpu.ids" | tail -3
for tg in 0 384 512; do
echo "== DM_WINO_Q_TARGET_WGS=$tg"
export DM_WINO_Q_TARGET_WGS=$tg
python3 tools/step_time.py --batch 256 --size 32
python3 tools/step_time.py --batch 64 --size 32
python3 tools/step_time.py --batch 32 --size 64
python3 tools/step_time.py --batch 8 --size 64
python3 tools/train_time.py --batch 64 --steps 30 --dropout 0.1 | tail -1
python3 tools/train_time.py --batch 16 --steps 30 | tail -1
done
