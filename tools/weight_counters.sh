#!/bin/bash
# SQ / GRBM counters and sustained clock / power of merge_bench builds (CT_PIVOT_WEIGHT variants), one at a time.
#   tools/weight_counters.sh <suffix> ...      for tools/merge_bench_<suffix>
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/weight_counters
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $R
for b in "$@"; do
  BIN=$R/tools/merge_bench_$b
  [ -x $BIN ] || continue
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY \
    -d $O/$b.p1 --output-format csv -- $BIN 32 4096 4096 4 "pivot V4 mult" > $O/$b.p1.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    -d $O/$b.p2 --output-format csv -- $BIN 32 4096 4096 4 "pivot V4 mult" > $O/$b.p2.log 2>&1
  # sustained alone with rocm-smi beside it
  $BIN 32 4096 4096 3000 "pivot V4 mult" > $O/$b.run.log 2>&1 &
  pid=$!
  : > $O/$b.smi
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' ' >> $O/$b.smi
    echo >> $O/$b.smi
    sleep 0.3
  done
  wait $pid
done
python3 - "$O" "$@" <<'PY'
import csv, glob, collections, re, sys
o, names = sys.argv[1], sys.argv[2:]
for b in names:
    vals = {}
    for p in ("p1", "p2"):
        for f in glob.glob(f"{o}/{b}.{p}/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(float); seen = set()
            for r in csv.DictReader(open(f)):
                if "merge_pivot" not in r["Kernel_Name"]:
                    continue
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
            for c, v in agg.items():
                vals[c] = v / max(1, len(seen))
    rows = []
    for line in open(f"{o}/{b}.smi"):
        m = re.search(r"\((\d+)Mhz\).*\(W\): ([\d.]+)", line)
        if m:
            rows.append((int(m.group(1)), float(m.group(2))))
    busy = [r for r in rows if r[1] > 500]
    mid = busy[2:-1] if len(busy) > 5 else busy
    clk = sum(r[0] for r in mid) / max(1, len(mid)); pw = sum(r[1] for r in mid) / max(1, len(mid))
    t = [l.strip() for l in open(f"{o}/{b}.run.log") if " med " in l]
    print(f"== {b}: sclk {clk:.0f} MHz, power {pw:.0f} W ({len(mid)} samples) | {t[0] if t else ''}")
    print("   " + "  ".join(f"{c} {v:.4g}" for c, v in sorted(vals.items())))
    if "SQ_LDS_IDX_ACTIVE" in vals and "GRBM_GUI_ACTIVE" in vals:
        print(f"   LDS array busy = IDX_ACTIVE / 256 CUs / (GUI_ACTIVE / 8) = {vals['SQ_LDS_IDX_ACTIVE'] / 256 / (vals['GRBM_GUI_ACTIVE'] / 8) * 100:.1f} %")
PY
