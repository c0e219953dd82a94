#!/bin/bash
# PMC passes of the benchmark step (DDIM-50 slice of the same denoise step as the DDPM-1000 headline; one sample() call):
#   bash tools/pmc_run.sh <tag>      -> gpurun_out/<tag>_{fetch,write,mfma,l2}/
# Separate runs per counter group, --kernel-trace only (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE takes 3).
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# PMC_CMD overrides the profiled command (final_profiles.sh uses it for the training step: tools/train_time.py)
CMD=${PMC_CMD:-"python3 bench.py --workload ddim50 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs --no-train"}
echo "pass fetch"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -o f -- $CMD > gpurun_out/${tag}_fetch.log 2>&1
echo "pass write"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -o w -- $CMD > gpurun_out/${tag}_write.log 2>&1
echo "pass mfma"; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_mfma -o m -- $CMD > gpurun_out/${tag}_mfma.log 2>&1
echo "pass l2"; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/${tag}_l2 -o l -- $CMD > gpurun_out/${tag}_l2.log 2>&1
for d in fetch write mfma l2; do rm -f gpurun_out/${tag}_$d/*kernel_trace.csv; ls -la gpurun_out/${tag}_$d | tail -3; done
