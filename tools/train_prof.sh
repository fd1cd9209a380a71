#!/bin/bash
# training parity tests, then per-kernel durations of the C3 training step (rocprofv3 --kernel-trace --stats),
# the standalone harness and the LDS atomic microbenchmark; outputs under gpurun_out/train_prof/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_prof
rm -rf $O && mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $O/st --output-format csv -- python3 bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline > $O/st.log 2>&1
grep "^{" $O/st.log > $O/bench_train_under_rocprof.json || true
python3 bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_train.json
cat $O/bench_train.json | cut -c1-400
cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats.csv")):
    if float(r["Percentage"]) > 0.5:
        print(r["Name"].split("<")[0][:40], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "pct", r["Percentage"])
PY
# harness executables are built on demand from tools/*.hip (see the header of each file)
[ -x ./tools/pairs_bench ] && ./tools/pairs_bench | tee $O/pairs_bench.log
[ -x ./tools/lds_atomic_rates ] && ./tools/lds_atomic_rates > $O/lds_atomic_rates.log
true
