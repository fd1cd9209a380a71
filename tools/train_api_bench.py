"""End-to-end check of the drop-in API on the C3 shape: train_icrf (per-channel optimizers, loss[c].backward() as in the
reference) over a device-resident StackDataset; prints ms per epoch (= per optimizer step, one batch per epoch)."""
import sys
import time

import torch
from torch.utils.data import DataLoader

sys.path.insert(0, ".")
from clair_torch_amd.common.enums import InterpMode  # noqa: E402
from clair_torch_amd.common.transforms import CastTo, Normalize  # noqa: E402
from clair_torch_amd.datasets import StackDataset, custom_collate, synthetic_exposure_stack  # noqa: E402
from clair_torch_amd.models import ICRFModelDirect  # noqa: E402
from clair_torch_amd.training import train_icrf  # noqa: E402

dev = torch.device("cuda:0")
n, size, epochs = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 12
unc = len(sys.argv) > 3 and sys.argv[3] == "unc"   # uncertainty-weighted loss with sigma = 0.05 x (derived in-kernel)
codes, exposures = synthetic_exposure_stack(n, 3, size, size, bits=16, stops_per_step=0.125, seed=1237, device=dev)
from clair_torch_amd.common.enums import MissingStdMode  # noqa: E402
ds = (StackDataset(codes, exposures, missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05, materialize_std=False)
      if unc else StackDataset(codes, exposures))   # tensors stay on the device: .to(device) in the loop is a no-op
loader = DataLoader(ds, batch_size=n, shuffle=False, collate_fn=custom_collate)
model = ICRFModelDirect(n_points=256, channels=3, interpolation_mode=InterpMode.LINEAR, initial_power=2.5).to(dev)
kw = dict(use_relative_linearity_loss=True, use_uncertainty_weighting=unc, patience=10 ** 6, alpha=10.0,
          exposure_ratio_threshold=0.25, gpu_transforms=[CastTo("float32"), Normalize(max_val=65535, min_val=0)], verbose=False)
train_icrf(loader, n, "cuda", model, epochs=3, **kw)   # warm-up (first step is the reference's dead step)
torch.cuda.synchronize()
t0 = time.perf_counter()
train_icrf(loader, n, "cuda", model, epochs=epochs, **kw)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"train_icrf public API{' (uncertainty-weighted)' if unc else ''}, {n}x{size}x{size}x3 uint16 on device: {el / epochs * 1e3:.2f} ms per epoch ({epochs / el:.1f} it/s)")
