#!/bin/bash
# default bench (C2 merge) under rocprofv3 --kernel-trace --stats, then plain; outputs under gpurun_out/merge_prof/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/merge_prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $O/m --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/m.log 2>&1
grep "^{" $O/m.log > $O/prof_bench_line.json
cp $(find $O/m -name "*kernel_stats.csv" | head -1) $O/merge_kernel_stats.csv
python3 bench.py > $O/bench_final.json 2>/dev/null
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/merge_kernel_stats.csv")))[:4]:
    print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
cut -c1-300 $O/bench_final.json
