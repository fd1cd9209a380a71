import sys, os, time, json, torch
sys.path.insert(0, os.getcwd())
from clair_torch_amd import ops
from clair_torch_amd.datasets import synthetic_exposure_stack
dev = torch.device("cuda:0")
codes, exposures = synthetic_exposure_stack(32, 3, 4096, 4096, bits=16, stops_per_step=0.25, seed=1236, device=dev)
t = torch.tensor(exposures, dtype=torch.float64, device=dev)
lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
out = {}
for mode in ("linear", "lookup", "catmull"):
    for ro in (False, True):
        for std_mode in ("multiplier", "none"):
            if std_mode == "none" and ro:
                continue
            kw = dict(lut=lut, interp=mode, gaussian_weight=True, std_mode=std_mode, std_value=0.05, reference_order=ro)
            for _ in range(3):
                ops.hdr_merge_batch(codes, t, **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                ops.hdr_merge_batch(codes, t, **kw)
            torch.cuda.synchronize()
            out[f"{mode} {'reference order' if ro else 'closed form'} std={std_mode}"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
print(json.dumps(out, indent=1))
