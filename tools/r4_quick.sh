set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "DM_PW_RT_FIRST=1" "DM_PW_RT_FIRST=0" "DM_PW_RT_FIRST=1 DM_PW_TARGET_WGS=512" "DM_PW_RT_FIRST=1 DM_PW_MIN_CHUNKS=4"; do
echo "== $cfg"
env $cfg python3 tools/step_time.py --batch 256 --size 32
env $cfg python3 tools/step_time.py --batch 64 --size 32
env $cfg python3 tools/step_time.py --batch 32 --size 64
env $cfg python3 tools/step_time.py --batch 8 --size 64
env $cfg python3 tools/train_time.py --batch 64 --steps 30 --dropout 0.1 | tail -1
done
