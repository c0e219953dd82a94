#!/bin/bash
# A/B of alternative builds of the library on one box:
#   PAT='^wino4<' bash tools/lib_ab.sh libA.so libB.so ...   (file names under the package dir; PAT = layer_report rows to sum)
PAT=${PAT:-'^pw<'}
for lib in "$@"; do
  export DM_LIB=$GRAFT_REPO_ROOT/diffusion-models_amd/$lib
  echo "== $lib"
  python tools/layer_report.py 2>/dev/null | grep -E "$PAT" | awk '{s+=$NF} END {print "matching layers, ms/fwd:", s}'
  python tools/step_time.py --batch 256 --size 32 2>/dev/null | tail -1
  python tools/step_time.py --batch 32 --size 64 2>/dev/null | tail -1
done
