#!/bin/bash
# SQ counters and HBM traffic of the C2 merge launch per input layout (bench.py --layout): why is the interleaved ingest slower?
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/pmc_layout
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for l in nchw nhwc; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    -d $O/$l.p1 --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --layout $l > $O/$l.p1.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
    -d $O/$l.p2 --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --layout $l > $O/$l.p2.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/$l.p3 --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --layout $l > $O/$l.p3.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/$l.p4 --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --layout $l > $O/$l.p4.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, collections, sys
o = sys.argv[1]
for l in ("nchw", "nhwc"):
    vals = {}
    for p in ("p1", "p2", "p3", "p4"):
        for f in glob.glob(f"{o}/{l}.{p}/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(float); seen = set()
            for r in csv.DictReader(open(f)):
                if "merge_pivot" not in r["Kernel_Name"] or int(r.get("Grid_Size", "0") or 0) < 100000:
                    continue
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
            for c, v in agg.items():
                vals[c] = v / max(1, len(seen))
    print(f"== {l}: " + "  ".join(f"{c} {v:.4g}" for c, v in sorted(vals.items())))
PY
