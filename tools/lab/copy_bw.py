import torch, time
dev = torch.device("cuda:0")
for mb in (403, 805, 101):
    n = mb * 1000 * 1000 // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"copy {mb} MB: {t*1e6:.1f} us  {2*n*4/t/1e12:.2f} TB/s (read+write)")
    # read-only: sum
    for _ in range(3): x.sum()
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): x.sum()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"sum  {mb} MB: {t*1e6:.1f} us  {n*4/t/1e12:.2f} TB/s (read)")
    for _ in range(3): y.zero_()
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): y.zero_()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"zero {mb} MB: {t*1e6:.1f} us  {n*4/t/1e12:.2f} TB/s (write)")
