set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
DM_BENCH_DEBUG=1 DM_BENCH_REHEARSE=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 1 --warmup 1 --batch 32 --workload ddim50 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/r4_rehearse2.json 2> gpurun_out/r4_rehearse2.err || (tail -20 gpurun_out/r4_rehearse2.err; false)
grep "host ms" gpurun_out/r4_rehearse2.err
tail -c 500 gpurun_out/r4_rehearse2.json
