import time, torch
dev = torch.device("cuda:0")
c, h, w = 3, 8192, 8192
mean = torch.rand((c, h, w), device=dev, dtype=torch.float64)
std = torch.rand((c, h, w), device=dev, dtype=torch.float32)
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("amin dim(1,2) f64", t(lambda: mean.amin(dim=(1, 2))))
print("amax dim(1,2) f64", t(lambda: mean.amax(dim=(1, 2))))
print("sum dim(1,2) f64", t(lambda: mean.sum(dim=(1, 2))))
print("aminmax view f64", t(lambda: torch.aminmax(mean.view(c, -1), dim=1)))
print("std.double()", t(lambda: std.double()))
print("std sum dtype f64", t(lambda: std.sum(dim=(1, 2), dtype=torch.float64)))
print("std aminmax", t(lambda: torch.aminmax(std.view(c, -1), dim=1)))
print("two-stage amin", t(lambda: mean.view(c, 4096, -1).amin(dim=2).amin(dim=1)))
