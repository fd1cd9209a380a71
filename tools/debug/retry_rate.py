"""Share of wavefronts of ct::merge_pivot_kernel that repeat the batch about the exact mean, per mode, on the C2 stack."""
import sys, os, ctypes, json, torch
sys.path.insert(0, os.getcwd())
from clair_torch_amd import ops, _native as nv
from clair_torch_amd.datasets import synthetic_exposure_stack
dev = torch.device("cuda:0")
codes, exposures = synthetic_exposure_stack(32, 3, 4096, 4096, bits=16, stops_per_step=0.25, seed=1236, device=dev)
t = torch.tensor(exposures, dtype=torch.float64, device=dev)
lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
lib = nv.load()
lib.ct_merge_set_retry_counter.argtypes = [ctypes.c_void_p]
waves = 3 * 4096 * 4096 // 256
out = {}
for mode in ("linear", "lookup"):
    buf = torch.zeros(1, dtype=torch.int64, device=dev)
    lib.ct_merge_set_retry_counter(ctypes.c_void_p(buf.data_ptr()))
    ops.hdr_merge_batch(codes, t, lut=lut, interp=mode, gaussian_weight=True, std_mode="multiplier", std_value=0.05, reference_order=False)
    torch.cuda.synchronize()
    lib.ct_merge_set_retry_counter(None)
    out[mode] = {"wavefronts": waves, "repeated": int(buf.item()), "share": round(int(buf.item()) / waves, 4)}
print(json.dumps(out))
