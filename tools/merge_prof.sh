#!/bin/bash
# default bench (C2 merge) under rocprofv3 --kernel-trace --stats, then plain; outputs under gpurun_out/merge_prof/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/merge_prof
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats -d $O/m --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/m.log 2>&1
grep "^{" $O/m.log > $O/prof_bench_line.json
cp $(find $O/m -name "*kernel_stats.csv" | head -1) $O/merge_kernel_stats.csv
# per-dispatch durations of the merge kernel in launch order: the 20 timed launches follow the 3 warm-up ones, the
# steady-state block is the last 100
python3 - <<PY > $O/trace_summary.txt
import csv, glob, json
rows = []
for f in glob.glob("$O/m/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "merge_pivot_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
us = [x for x in us if x > 100.0]  # drop bench.py's library-initialisation merge of an 8 x 3 x 8 x 8 stack (a few us)
line = json.loads(open("$O/prof_bench_line.json").read().strip().splitlines()[-1])
w, k = line["warmup"], line["steps"]
timed, steady = us[w:w + k], us[-100:]
print(f"rocprofv3 --kernel-trace, ct::merge_pivot_kernel: {len(us)} dispatches")
print(f"timed region (dispatches {w + 1}..{w + k}): mean {sum(timed) / len(timed):.1f} us   bench.py kernel_ms {line['roofline']['kernel_ms'] * 1e3:.1f} us")
print(f"steady-state block (last 100): mean {sum(steady) / len(steady):.1f} us   bench.py steady_state.kernel_ms {line['roofline']['steady_state']['kernel_ms'] * 1e3:.1f} us")
print("all dispatches in order (us):", " ".join(f"{x:.0f}" for x in us))
PY
cat $O/trace_summary.txt | head -3
python3 bench.py > $O/bench_final.json 2>/dev/null
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/merge_kernel_stats.csv")))[:4]:
    print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
cut -c1-300 $O/bench_final.json
