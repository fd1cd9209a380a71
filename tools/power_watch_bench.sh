#!/bin/bash
# rocm-smi shader clock / socket power sampled while bench.py workloads run long timed regions
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/power_watch_bench
rm -rf $O && mkdir -p $O
cd $R
run() {  # tag, bench args...
  local tag=$1; shift
  python3 bench.py "$@" --no-cpu-baseline > $O/$tag.json 2>$O/$tag.err &
  local pid=$!
  : > $O/$tag.smi
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' ' >> $O/$tag.smi
    echo >> $O/$tag.smi
    sleep 0.3
  done
  wait $pid
  python3 - "$tag" "$O" <<'PY'
import json, re, sys
tag, o = sys.argv[1], sys.argv[2]
rows = []
for line in open(f"{o}/{tag}.smi"):
    m = re.search(r"\((\d+)Mhz\).*\(W\): ([\d.]+)", line)
    if m:
        rows.append((int(m.group(1)), float(m.group(2))))
top = max(r[1] for r in rows)
busy = [r for r in rows if r[1] > 0.9 * top]
clk = sum(r[0] for r in busy) / len(busy); pw = sum(r[1] for r in busy) / len(busy)
try:
    j = json.loads(open(f"{o}/{tag}.json").read().strip().splitlines()[-1])
    t = f"ms_per_step {j['ms_per_step']} frac {j['roofline']['frac']}"
except Exception as e:
    t = f"no bench line ({e})"
print(f"{tag:12s} sclk {clk:6.0f} MHz  power {pw:6.0f} W  ({len(busy)} samples at > 90 % of the peak reading) | {t}", flush=True)
PY
}
run merge --steps 4000 --warmup 50
run train --workload train --steps 350 --warmup 5
run linearize --workload linearize --steps 4000 --warmup 50
run video --workload video --steps 6000 --warmup 50
run flatfield --workload flatfield --steps 6000 --warmup 50
