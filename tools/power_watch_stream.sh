#!/bin/bash
# sustained, isolated runs of the streaming yardsticks (8 and 16 bytes per lane and load) with clock / power sampling
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/power_watch_stream
rm -rf $O && mkdir -p $O
cd $R
run() {
  local tag=$1 rounds=$2 filt=$3
  ./tools/merge_bench 32 4096 4096 $rounds "$filt" > $O/$tag.log 2>&1 &
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
print(f"{tag:16s} sclk {clk:6.0f} MHz  power {pw:6.0f} W  ({len(mid)} samples) | {t[0].strip() if t else 'no timing'}", flush=True)
PY
}
for pass in 1 2; do
  run stream_v4_$pass 5000 "stream V4"
  run stream_v8_$pass 5000 "stream V8"
done
