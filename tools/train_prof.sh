#!/bin/bash
# training parity tests, then per-kernel durations of the C3 training step (rocprofv3 --kernel-trace --stats)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_prof
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $O/st --output-format csv -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $O/st.log 2>&1
grep "^{" $O/st.log | cut -c1-200
python3 - <<PY
import csv, glob
for f in glob.glob("$O/st/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print(r["Name"].split("<")[0][:40], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
