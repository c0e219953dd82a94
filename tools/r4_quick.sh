set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/conv_stamps.py --batch 256 2> gpurun_out/r4_conv_stamps_b256.txt > /dev/null || true
grep -c STAMPS gpurun_out/r4_conv_stamps_b256.txt
python3 tools/layer_report.py --batch 256 --size 32 --bounds > gpurun_out/r4_layer_report_b256_32.txt 2>&1
python3 tools/layer_report.py --batch 8 --size 64 --bounds > gpurun_out/r4_layer_report_b8_64.txt 2>&1
tail -12 gpurun_out/r4_layer_report_b8_64.txt
