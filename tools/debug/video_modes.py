import sys, os, time, json, torch
sys.path.insert(0, os.getcwd())
from clair_torch_amd import ops
dev = torch.device("cuda:0")
b, c, h, w = 32, 3, 1080, 1920
frames = torch.randint(0, 65536, (b, c, h, w), device=dev, dtype=torch.int32).to(torch.uint16)
lut = torch.stack([torch.linspace(0, 1, 256) ** p for p in (2.2, 2.4, 2.6)]).to(dev)
mean = torch.zeros((c, h, w), device=dev); m2 = torch.zeros_like(mean)
out = {}
for mode in (None, "lookup", "linear", "catmull"):
    for bs in (16, 32):
        f = frames[:bs]
        for _ in range(10):
            ops.video_stats_batch(f, mean, m2, 0, lut=None if mode is None else lut, interp=mode)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            ops.video_stats_batch(f, mean, m2, 32, lut=None if mode is None else lut, interp=mode)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        gb = c * h * w * (bs * 2 + 16) / 1e9
        out[f"{mode} B={bs}"] = {"ms": round(ms, 4), "TB/s": round(gb / ms, 3)}
print(json.dumps(out, indent=1))
