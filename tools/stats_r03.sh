#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the training step (C3) and the resident linearization launch (C4), round 3
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stats_r03
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $O/train --output-format csv -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/train.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/lin --output-format csv -- python3 bench.py --workload linearize --steps 50 --warmup 5 --no-cpu-baseline > $O/lin.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/lin_bgr --output-format csv -- python3 bench.py --workload linearize --steps 50 --warmup 5 --no-cpu-baseline --layout nhwc_bgr > $O/lin_bgr.log 2>&1
for w in train lin lin_bgr; do
  cp $(find $O/$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
  python3 - <<PY
import csv
print("== $w")
for r in csv.DictReader(open("$O/${w}_kernel_stats.csv")):
    if float(r["Percentage"]) > 1.0:
        print("  ", r["Name"][:80], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
done
