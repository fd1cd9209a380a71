import sys, os, time, torch, cProfile, pstats
sys.path.insert(0, os.getcwd())
from torch.utils.data import DataLoader
from clair_torch_amd.common.enums import InterpMode, MissingStdMode
from clair_torch_amd.common.transforms import CastTo, Normalize
from clair_torch_amd.datasets import StackDataset, custom_collate
from clair_torch_amd.inference import compute_hdr_image
from clair_torch_amd.models import ICRFModelDirect
from clair_torch_amd.training.losses import gaussian_value_weights
dev = torch.device("cuda:0")
host = torch.zeros((32, 3, 4096, 4096), dtype=torch.uint16).pin_memory()
t = [0.001 * 2 ** (k / 4) for k in range(32)]
lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)])
model = ICRFModelDirect(icrf=lut, interpolation_mode=InterpMode.LINEAR).to(dev)
ds = StackDataset(host, t, missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05, materialize_std=False)
tf = [CastTo("float32"), Normalize(max_val=65535, min_val=0)]
loader = DataLoader(ds, batch_size=4, shuffle=False, collate_fn=custom_collate)
for _ in range(2):
    compute_hdr_image(loader, dev, model, weight_fn=gaussian_value_weights, gpu_transforms=tf)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
compute_hdr_image(loader, dev, model, weight_fn=gaussian_value_weights, gpu_transforms=tf)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
pr.disable()
print("host %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
