set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in ${BATCHES:-64 16}; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_tr$B -o t -- python3 tools/train_time.py --batch $B --warmup 1 --steps 3 > gpurun_out/r4_tr$B.log 2>&1
python3 tools/trace_tail.py gpurun_out/r4_tr$B/*kernel_trace.csv mse_loss > gpurun_out/r4_train_seq_b$B.txt || true
rm -f gpurun_out/r4_tr$B/*kernel_trace.csv
tail -2 gpurun_out/r4_tr$B.log
tail -1 gpurun_out/r4_train_seq_b$B.txt
done
python3 tools/train_time.py --batch 64 --steps 20 --host
