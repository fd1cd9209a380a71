#!/bin/bash
# samples rocm-smi clocks / power while the C2 merge runs back to back for a few seconds
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/clock_watch
rm -rf $O && mkdir -p $O
cd $R
python3 - > $O/run.log 2>&1 <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
import bench
from clair_torch_amd import ops
from clair_torch_amd.datasets import synthetic_exposure_stack
dev = torch.device("cuda:0")
codes, exposures = synthetic_exposure_stack(32, 3, 4096, 4096, bits=16, stops_per_step=0.25, seed=1236, device=dev)
lut = bench.make_lut(dev)
t_dev = torch.tensor(exposures, dtype=torch.float64, device=dev)
kw = dict(lut=lut, interp="linear", gaussian_weight=True, std_mode="multiplier", std_value=0.05)
torch.cuda.synchronize()
print("start", time.time(), flush=True)
t_end = time.time() + 8.0
n = 0
while time.time() < t_end:
    for _ in range(200):
        ops.hdr_merge_batch(codes, t_dev, **kw)
    torch.cuda.synchronize()
    n += 200
print("end", time.time(), n, flush=True)
PY
PID=$!
for i in $(seq 1 40); do
  date +%s.%N >> $O/smi.log
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" >> $O/smi.log
  kill -0 $PID 2>/dev/null || break
  sleep 0.5
done
wait $PID
cat $O/run.log | tail -3
