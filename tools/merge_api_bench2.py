"""compute_hdr_image through the public API on the C2 stack, device-resident dataset: batch_size 32 / 8 / 4 (the reference's default)."""
import sys, os, time, json, torch
sys.path.insert(0, os.getcwd())
from torch.utils.data import DataLoader
from clair_torch_amd.common.enums import InterpMode, MissingStdMode
from clair_torch_amd.common.transforms import CastTo, Normalize
from clair_torch_amd.datasets import StackDataset, custom_collate, synthetic_exposure_stack
from clair_torch_amd.inference import compute_hdr_image
from clair_torch_amd.models import ICRFModelDirect
from clair_torch_amd.training.losses import gaussian_value_weights
dev = torch.device("cuda:0")
codes, exposures = synthetic_exposure_stack(32, 3, 4096, 4096, bits=16, stops_per_step=0.25, seed=1236, device=dev)
lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)])
mode = (sys.argv[1] if len(sys.argv) > 1 else "linear").upper()   # linear | lookup | catmull
model = ICRFModelDirect(icrf=lut, interpolation_mode=InterpMode[mode]).to(dev)
ds = StackDataset(codes, exposures, missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05, materialize_std=False)
tf = [CastTo("float32"), Normalize(max_val=65535, min_val=0)]
out = {"interp": mode.lower()}
for bs in (32, 8, 4):
    loader = DataLoader(ds, batch_size=bs, shuffle=False, collate_fn=custom_collate)
    for _ in range(3):
        compute_hdr_image(loader, dev, model, weight_fn=gaussian_value_weights, gpu_transforms=tf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        mean, std = compute_hdr_image(loader, dev, model, weight_fn=gaussian_value_weights, gpu_transforms=tf)
    torch.cuda.synchronize()
    out[f"batch_size_{bs}_ms_per_call"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
print(json.dumps(out))
