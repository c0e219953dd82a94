#!/bin/bash
# A/B of the second-stream fork (DM_PAR) at the batches of configs 3 / 1:  bash tools/par_ab.sh  -> gpurun_out/par_ab.txt
out=gpurun_out/par_ab.txt; : > $out
for cfg in "8 64" "32 64" "64 32" "256 32"; do
  set -- $cfg
  for par in 0 1; do
    echo -n "DM_PAR=$par " >> $out
    DM_PAR=$par python3 tools/step_time.py --batch $1 --size $2 --steps 100 2>/dev/null | tail -1 >> $out
  done
done
cat $out
