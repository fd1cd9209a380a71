"""Per-launch duration of the C2 merge over a long run of back-to-back launches (after setup, and again after an idle
pause): shows the clock / power transient a 20-launch timed region sits in.  Run on the GPU box; prints one line per
10 launches."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from clair_torch_amd import ops  # noqa: E402
from clair_torch_amd.datasets import synthetic_exposure_stack  # noqa: E402

dev = torch.device("cuda:0")
codes, exposures = synthetic_exposure_stack(32, 3, 4096, 4096, bits=16, stops_per_step=0.25, seed=1236, device=dev)
lut = bench.make_lut(dev)
t_dev = torch.tensor(exposures, dtype=torch.float64, device=dev)
kw = dict(lut=lut, interp="linear", gaussian_weight=True, std_mode="multiplier", std_value=0.05)


def series(n, label):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record()
        ops.hdr_merge_batch(codes, t_dev, **kw)
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    print(label, "first 10:", " ".join(f"{x:.3f}" for x in ms[:10]))
    for k in range(10, n, 10):
        chunk = ms[k:k + 10]
        print(f"{label} launches {k:4d}-{k + len(chunk) - 1:4d}: mean {sum(chunk) / len(chunk):.3f} min {min(chunk):.3f} max {max(chunk):.3f}")


series(400, "after setup")
time.sleep(3.0)
series(200, "after 3 s idle")
