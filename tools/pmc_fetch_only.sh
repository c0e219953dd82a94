#!/bin/bash
# One FETCH_SIZE pass of the DDIM-50 slice (A/B runs of a traffic change):  bash tools/pmc_fetch_only.sh <tag>
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --workload ddim50 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs --no-train"
echo "pass fetch $tag"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -o f -- $CMD > gpurun_out/${tag}_fetch.log 2>&1
rm -f gpurun_out/${tag}_fetch/*kernel_trace.csv
python3 - <<PY
import csv, glob, re
from collections import defaultdict
t, c = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/${tag}_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", r["Kernel_Name"])).replace("dm::", "")
        t[n] += float(r["Counter_Value"]); c[n] += 1
for n in sorted(t, key=lambda k: -t[k])[:8]:
    print(f"{n:45s} launches {c[n]:6d}  fetch MB/launch (x2 corrected) {t[n] / c[n] * 2048 / 1e6:8.2f}")
PY
