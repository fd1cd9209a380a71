"""End-to-end check of the drop-in API on the C2 shape: compute_hdr_image over a device-resident StackDataset of raw
uint16 codes (CastTo + Normalize folded into the kernel, sigma = 0.05 x derived in-kernel); prints ms per call."""
import sys
import time

import torch
from torch.utils.data import DataLoader

sys.path.insert(0, ".")
from clair_torch_amd.common.enums import InterpMode, MissingStdMode  # noqa: E402
from clair_torch_amd.common.transforms import CastTo, Normalize  # noqa: E402
from clair_torch_amd.datasets import StackDataset, custom_collate, synthetic_exposure_stack  # noqa: E402
from clair_torch_amd.inference import compute_hdr_image  # noqa: E402
from clair_torch_amd.models import ICRFModelDirect  # noqa: E402
from clair_torch_amd.training.losses import gaussian_value_weights  # noqa: E402

dev = torch.device("cuda:0")
n, size = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
codes, exposures = synthetic_exposure_stack(n, 3, size, size, bits=16, stops_per_step=0.25, seed=1236, device=dev)
ds = StackDataset(codes, exposures, missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05, materialize_std=False)
model = ICRFModelDirect(n_points=256, channels=3, interpolation_mode=InterpMode.LINEAR, initial_power=2.4).to(dev)
tf = [CastTo("float32"), Normalize(max_val=65535, min_val=0)]
for batch in (n, 8):
    loader = DataLoader(ds, batch_size=batch, shuffle=False, collate_fn=custom_collate)
    compute_hdr_image(loader, "cuda", model, weight_fn=gaussian_value_weights, gpu_transforms=tf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        mean, std = compute_hdr_image(loader, "cuda", model, weight_fn=gaussian_value_weights, gpu_transforms=tf)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print(f"compute_hdr_image public API, {n}x{size}x{size}x3 uint16 on device, batch_size {batch}: {el * 1e3:.2f} ms "
          f"({size * size / el / 1e6:.0f} MPix/s merged)")
