#!/bin/bash
# SQ counters of the default bench (C2 merge kernel): per-dispatch averages
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_merge
rm -rf $O && mkdir -p $O
cd $R
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  -d $O/p1 --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM \
  -d $O/p2 --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_INSTS_SMEM \
  -d $O/p3 --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/p3.log 2>&1 || true
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key); cnt[k] += 1
        for k in agg:
            if "merge_" in k and "kernel" in k:
                print(p, k[-40:], "dispatches", cnt[k], {c: "%.4g" % (v / cnt[k]) for c, v in agg[k].items()})
PY
