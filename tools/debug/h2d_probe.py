import torch, time
dev = torch.device("cuda:0")
for mb in (64, 400, 3200):
    n = mb * 1024 * 1024 // 2
    h = torch.empty(n, dtype=torch.uint16).pin_memory()
    d = torch.empty(n, dtype=torch.uint16, device=dev)
    for direction in ("h2d", "d2h"):
        for _ in range(2):
            (d.copy_(h, non_blocking=True) if direction == "h2d" else h.copy_(d, non_blocking=True))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            (d.copy_(h, non_blocking=True) if direction == "h2d" else h.copy_(d, non_blocking=True))
        torch.cuda.synchronize()
        print(mb, "MB", direction, round(n * 2 * 5 / (time.perf_counter() - t0) / 1e9, 1), "GB/s", flush=True)
# two streams, halves of the buffer
n = 3200 * 1024 * 1024 // 2
h = torch.empty(n, dtype=torch.uint16).pin_memory(); d = torch.empty(n, dtype=torch.uint16, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1): d[: n // 2].copy_(h[: n // 2], non_blocking=True)
    with torch.cuda.stream(s2): d[n // 2:].copy_(h[n // 2:], non_blocking=True)
torch.cuda.synchronize()
print("3200 MB h2d on two streams", round(n * 2 * 5 / (time.perf_counter() - t0) / 1e9, 1), "GB/s")
