cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -x -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -15 > gpurun_out/r4_fullsuite.txt
tail -5 gpurun_out/r4_fullsuite.txt
