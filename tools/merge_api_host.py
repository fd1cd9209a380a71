"""compute_hdr_image on the C2 stack held in PINNED HOST memory (what a file-backed Dataset would hand over): the
PCIe-inclusive rate of the merge.  The boundary itself takes device pointers; this is the public API in front of it."""
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
host = torch.empty(codes.shape, dtype=codes.dtype).pin_memory()
host.copy_(codes)
del codes
torch.cuda.empty_cache()
lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)])
model = ICRFModelDirect(icrf=lut, interpolation_mode=InterpMode.LINEAR).to(dev)
ds = StackDataset(host, exposures, missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05, materialize_std=False)
tf = [CastTo("float32"), Normalize(max_val=65535, min_val=0)]
out = {"stack": "32x3x4096x4096 uint16 in pinned host memory (3.22 GB)"}
for bs in (32, 4):
    loader = DataLoader(ds, batch_size=bs, shuffle=False, collate_fn=custom_collate)
    for _ in range(2):
        compute_hdr_image(loader, dev, model, weight_fn=gaussian_value_weights, gpu_transforms=tf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        mean, std = compute_hdr_image(loader, dev, model, weight_fn=gaussian_value_weights, gpu_transforms=tf)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    out[f"batch_size_{bs}"] = {"ms_per_call": round(ms, 2), "MPix_per_s": round(4096 * 4096 / ms / 1e3, 1),
                               "host_to_device_GB_per_s": round(host.numel() * 2 / ms / 1e6, 1)}
print(json.dumps(out))
