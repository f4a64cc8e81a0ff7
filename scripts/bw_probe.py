"""Practical HBM bandwidth on this box (torch copy / read-only reduction), for context next to the roofline numbers."""
import torch, time
dev = "cuda:0"
for mb in (64, 256, 1024, 3072):
    n = mb * 1024 * 1024 // 4
    x = torch.rand(n, device=dev); y = torch.empty_like(x)
    for name, fn, bytes_ in (("copy", lambda: y.copy_(x), 8 * n), ("read(sum)", lambda: x.sum(), 4 * n), ("write(fill)", lambda: y.fill_(1.0), 4 * n)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mb:5d} MiB {name:12s} {ms*1e3:9.1f} us  {bytes_/ms/1e9:7.2f} TB/s".replace("TB/s", "GB/ms = TB/s"))
