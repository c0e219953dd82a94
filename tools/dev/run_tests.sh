# development aid: bash tools/dev/run_tests.sh <pytest args...>   (on the GPU box, from the repo root)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest "$@" 2>&1 | grep -v "amdgpu.ids" | tail -15
