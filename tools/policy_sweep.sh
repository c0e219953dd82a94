#!/bin/bash
# Step time of a small-batch shape under different tiling-policy knobs:  bash tools/policy_sweep.sh [batch] [size]
B=${1:-8}; S=${2:-64}
run() { echo -n "$* : "; env "$@" python tools/step_time.py --batch $B --size $S | tail -1; }
run DM_X=0
run DM_CONV_TARGET_WGS=256
run DM_CONV_TARGET_WGS=128
run DM_CONV_MAX_SPLITS=4
run DM_CONV_MAX_SPLITS=2
run DM_CONV_MAX_SPLITS=1
run DM_WINO_TARGET_WGS=128
run DM_WINO_TARGET_WGS=64
run DM_WINO_TARGET_WGS=512
run DM_WINO_MIN_CHUNKS=4
run DM_WINO_MIN_CHUNKS=16
run DM_WINO4_MIN_WGS=32
run DM_WINO4_MIN_WGS=64 DM_WINO4_MIN_K=8
