import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from torch.utils.data import DataLoader
from clair_torch_amd.common.enums import MissingStdMode
from clair_torch_amd.datasets import StackDataset, custom_collate
dev = torch.device("cuda:0")
host = torch.empty((32, 3, 4096, 4096), dtype=torch.uint16).pin_memory()
host.random_(0, 65535) if False else None
t = [0.001 * 2 ** (k / 4) for k in range(32)]
ds = StackDataset(host, t, missing_std_mode=MissingStdMode.MULTIPLIER, missing_std_value=0.05, materialize_std=False)
for bs in (32, 4):
    loader = DataLoader(ds, batch_size=bs, shuffle=False, collate_fn=custom_collate)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = []
        for _, val, std, meta in loader:
            t1 = time.perf_counter()
            outs.append(val.to(dev, non_blocking=True))
        t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print("bs", bs, "pinned", val.is_pinned(), "contig", val.is_contiguous(), "host loop %.1f ms, total %.1f ms -> %.1f GB/s" % ((t2 - t0) * 1e3, (t3 - t0) * 1e3, host.numel() * 2 / (t3 - t0) / 1e9), flush=True)
        del outs
