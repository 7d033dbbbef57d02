"""Lab: achievable HBM bandwidth for the traffic shapes of the pointwise convs, with trivial torch kernels."""
import time
import torch
dev = torch.device("cuda:0")
M = 1088 * 56 * 56
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
x64 = torch.randn(M, 64, device=dev); o128 = torch.empty(M, 128, device=dev); o64 = torch.empty(M, 64, device=dev)
x128 = torch.randn(M, 128, device=dev)
def cat2(): torch.cat([x64, x64], 1, out=o128)
def copy(): o64.copy_(x64)
def half(): o64.copy_(x128[:, :64])
def add(): torch.add(x128, 1.0, out=o128)
for name, fn, b in (("copy 64->64 (1:1)", copy, 2 * M * 64 * 4), ("cat 64->128 (1:2)", cat2, 3 * M * 64 * 4),
                    ("x128+1 -> 128 (1:1)", add, 2 * M * 128 * 4), ("slice 128[:64]->64", half, 2 * M * 64 * 4)):
    dt = t(fn)
    print(f"{name:24s} {dt*1e6:8.1f} us  {b/dt/1e12:5.2f} TB/s")
def fill(): o128.fill_(1.5)
def rd(): return x128.sum()
for name, fn, b in (("fill 128 (write only)", fill, M * 128 * 4), ("sum 128 (read only)", rd, M * 128 * 4)):
    dt = t(fn)
    print(f"{name:24s} {dt*1e6:8.1f} us  {b/dt/1e12:5.2f} TB/s")
