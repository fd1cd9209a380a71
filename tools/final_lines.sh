#!/bin/bash
# The round's bench lines on ONE box with the tree as committed: every workload bench.py knows, one JSON each under
# gpurun_out/final/ (copied to profiles/r03_final_*.json), then the HBM traffic of the training step on an interleaved
# stack (separate PMC passes).  Join with && so that a failing line stops the battery.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O
cd $R
run() { name=$1; shift; python3 bench.py "$@" > $O/$name.json 2> $O/$name.err; python3 - <<PY
import json
d = json.loads(open("$O/$name.json").read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print("%-28s %10.3f %-10s %8.3f ms  frac %s" % ("$name", d["value"], d["unit"][:10], d["ms_per_step"], r.get("frac")))
PY
}
run merge_c2 --gpus 1 --steps 20 --warmup 5
run merge_c2_nhwc --steps 20 --warmup 5 --no-cpu-baseline --layout nhwc
run merge_c2_nhwc_bgr --steps 20 --warmup 5 --no-cpu-baseline --layout nhwc_bgr
run merge_c2_nhwc_bgr_out_as_input --steps 20 --warmup 5 --no-cpu-baseline --layout nhwc_bgr --out-layout input
run merge_c2_max4095 --steps 20 --warmup 5 --no-cpu-baseline --max-code 4095
run merge_c2_f32 --steps 20 --warmup 5 --no-cpu-baseline --input f32
run merge_c2_lookup --steps 20 --warmup 5 --no-cpu-baseline --interp lookup
run merge_c2_catmull --steps 20 --warmup 5 --no-cpu-baseline --interp catmull
run merge_c2_catmull_mean_only --steps 20 --warmup 5 --no-cpu-baseline --interp catmull --std none
run merge_c2_linear_mean_only --steps 20 --warmup 5 --no-cpu-baseline --std none
run merge_c5_strong_n1 --steps 10 --warmup 3 --no-cpu-baseline --scaling strong
run linearize_c4 --workload linearize --steps 200 --warmup 20
run linearize_c4_bgr --workload linearize --steps 200 --warmup 20 --no-cpu-baseline --layout nhwc_bgr
run linearize_c4_streamed --workload linearize --streamed --no-cpu-baseline
run train_c3 --workload train --steps 30 --warmup 5
run train_c3_nhwc_bgr --workload train --steps 30 --warmup 5 --no-cpu-baseline --layout nhwc_bgr
run train_c3_max4095 --workload train --steps 30 --warmup 5 --no-cpu-baseline --max-code 4095
run train_c3_through_api --workload train --through-api --steps 100 --warmup 5 --no-cpu-baseline
run video --workload video --no-cpu-baseline
run flatfield --workload flatfield --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp && cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_train_nhwc_$c --output-format csv -- python3 bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline --layout nhwc > $O/pmc_train_nhwc_$c.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/pmc_train_nhwc_%s/**/*counter_collection.csv" % c, recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "ct::" not in k: continue
            agg[k] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key); cnt[k] += 1
        for k in agg:
            print("train nhwc", c, k[-60:], "dispatches", cnt[k], "KB per dispatch %.1f" % (agg[k] / cnt[k]))
PY
