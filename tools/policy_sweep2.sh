#!/bin/bash
# Dispatch-threshold sweep of the 3x3 / upsample / 1x1 kernels at a given shape:  bash tools/policy_sweep2.sh [batch] [size]
B=${1:-128}; S=${2:-32}
run() { echo -n "$* : "; env "$@" python tools/step_time.py --batch $B --size $S 2>/dev/null | tail -1; }
run DM_X=0
run DM_WINO4_MIN_WGS=100
run DM_WINO4_MIN_WGS=128
run DM_WINO4_MIN_WGS=300
run DM_WINO4_MIN_K=8
run DM_WINO4_MIN_K=16
run DM_UPWINO_MIN_WGS=64
run DM_UPWINO_MIN_WGS=256
run DM_WINO_TARGET_WGS=128
run DM_WINO_TARGET_WGS=512
run DM_WINO4_TARGET_WGS=128
run DM_WINO4_TARGET_WGS=512
run DM_PW_TARGET_WGS=128
run DM_PW_TARGET_WGS=512
run DM_UPWINO_TARGET_WGS=128
run DM_UPWINO_TARGET_WGS=512
