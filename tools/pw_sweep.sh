#!/bin/bash
# 1x1 GEMM kernel policy sweep:  bash tools/pw_sweep.sh [batch] [size]
B=${1:-256}; S=${2:-32}
run() { echo -n "$* : "; env "$@" python tools/step_time.py --batch $B --size $S 2>/dev/null | tail -1; }
run DM_X=0
run DM_PW_TARGET_WGS=256
run DM_PW_TARGET_WGS=1024
run DM_PW_MIN_CHUNKS=4
run DM_PW_MIN_CHUNKS=16
run DM_PW_TARGET_WGS=256 DM_PW_MIN_CHUNKS=16
run DM_PW_TARGET_WGS=1024 DM_PW_MIN_CHUNKS=4
run DM_NO_PW=1
