set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in 0 8; do
echo "== DM_WINO4_ROUND_K=$k"
DM_WINO4_ROUND_K=$k python3 tools/step_time.py --batch 256 --size 32
DM_WINO4_ROUND_K=$k python3 tools/step_time.py --batch 64 --size 32
DM_WINO4_ROUND_K=$k python3 tools/step_time.py --batch 32 --size 64
DM_WINO4_ROUND_K=$k python3 tools/step_time.py --batch 8 --size 64
done
python3 tools/layer_report.py --batch 8 --size 64 > gpurun_out/r4_layer_report_b8_64.txt 2>&1
python3 tools/layer_report.py --batch 256 --size 32 > gpurun_out/r4_layer_report_b256_32.txt 2>&1
tail -5 gpurun_out/r4_layer_report_b8_64.txt
