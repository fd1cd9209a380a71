#!/bin/bash
# HBM traffic (TCC FETCH_SIZE / WRITE_SIZE, separate passes as the guide prescribes) of the linearize and training benches
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_traffic
rm -rf $O && mkdir -p $O
cd $R
for w in linearize train; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/${w}_$c --output-format csv -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/${w}_$c.log 2>&1
  done
done
python3 - <<PY
import csv, glob, collections
for w in ("linearize", "train"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (w, c), recursive=True):
            agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0]
                if "ct::" not in k: continue
                agg[k] += float(r["Counter_Value"])
                key = (r["Dispatch_Id"], k)
                if key not in seen:
                    seen.add(key); cnt[k] += 1
            for k in agg:
                print(w, c, k[-60:], "dispatches", cnt[k], "KB per dispatch %.1f" % (agg[k] / cnt[k]))
PY
