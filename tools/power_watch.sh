#!/bin/bash
# Sustained runs (a few seconds each) of the merge harness variants, one at a time, with rocm-smi sampled alongside:
# shader clock and socket power each variant settles at, next to its median launch time.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/power_watch
rm -rf $O && mkdir -p $O
cd $R
cp tools/merge_bench $O/merge_bench
run() {  # name, rounds, filter
  local tag=$1 rounds=$2 filt=$3
  $O/merge_bench 32 4096 4096 $rounds "$filt" > $O/$tag.log 2>&1 &
  local pid=$!
  : > $O/$tag.smi
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' ' >> $O/$tag.smi
    echo >> $O/$tag.smi
    sleep 0.3
  done
  wait $pid
  python3 - "$tag" "$O" <<'PY'
import re, sys
tag, o = sys.argv[1], sys.argv[2]
rows = []
for line in open(f"{o}/{tag}.smi"):
    m = re.search(r"\((\d+)Mhz\).*\(W\): ([\d.]+)", line)
    if m:
        rows.append((int(m.group(1)), float(m.group(2))))
busy = [r for r in rows if r[1] > 500]
mid = busy[2:-1] if len(busy) > 5 else busy
t = [l for l in open(f"{o}/{tag}.log") if " med " in l]
clk = sum(r[0] for r in mid) / max(1, len(mid)); pw = sum(r[1] for r in mid) / max(1, len(mid))
print(f"{tag:28s} sclk {clk:6.0f} MHz  power {pw:6.0f} W  ({len(mid)} samples) | {t[0].strip() if t else 'no timing'}")
PY
}
run pivot_mult 3500 "pivot V4 mult"
run compute_only 3500 "compute-only"
run pivot_nostd 4500 "pivot V4 nostd"
run stream_v4 5000 "stream V4"
run f64_mult 2800 "f64 V4 PF2 mult"
