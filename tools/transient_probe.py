"""Per-launch time of the first 60 C2 merges after set-up, for two orders of the set-up work (no extra GPU work in either):
  a) as bench.py does it today: generate the stack, then the first step() also loads the library, runs the host-side
     constant proofs and the occupancy query (the GPU idles meanwhile);
  b) library initialised first (one merge of an 8 x 3 x 8 x 8 stack), then the stack is generated and the launches follow
     the generation kernels without a gap.
usage: python tools/transient_probe.py {a|b}"""
import sys, time
import torch
sys.path.insert(0, ".")
from clair_torch_amd import ops
from clair_torch_amd.datasets import synthetic_exposure_stack

mode = sys.argv[1] if len(sys.argv) > 1 else "a"
dev = torch.device("cuda:0")
lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
kw = dict(lut=lut, interp="linear", gaussian_weight=True, std_mode="multiplier", std_value=0.05)
if mode == "b":
    c0, e0 = synthetic_exposure_stack(8, 3, 8, 8, bits=16, stops_per_step=0.25, seed=1, device=dev)
    ops.hdr_merge_batch(c0, torch.tensor(e0, dtype=torch.float64, device=dev), **kw)
    torch.cuda.synchronize()
codes, exposures = synthetic_exposure_stack(32, 3, 4096, 4096, bits=16, stops_per_step=0.25, seed=1236, device=dev)
t_dev = torch.tensor(exposures, dtype=torch.float64, device=dev)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
for a, b in ev:
    a.record()
    ops.hdr_merge_batch(codes, t_dev, **kw)
    b.record()
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in ev]
print(mode, "launch 1..60 ms:", " ".join(f"{x:.3f}" for x in ms))
print(mode, f"mean of launches 6..25: {sum(ms[5:25]) / 20:.4f} ms   launches 41..60: {sum(ms[40:60]) / 20:.4f} ms")
