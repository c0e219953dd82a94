set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_hip_model.py tests/test_inception.py tests/test_hip_ops.py tests/test_hip_forced_dispatch.py -x -q 2>&1 | grep -v "amdgpu.ids" | tail -5
python3 tools/vae_time.py 2>&1 | tail -3
DM_BENCH_REHEARSE=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 1 --warmup 1 --batch 32 --workload ddim50 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/r4_rehearse2.json 2> gpurun_out/r4_rehearse2.err || (tail -20 gpurun_out/r4_rehearse2.err; false)
python3 -c "
import json; d=json.load(open('gpurun_out/r4_rehearse2.json')); print({k:d[k] for k in ('value','n_gpus')}); print(d['train_step'])"
