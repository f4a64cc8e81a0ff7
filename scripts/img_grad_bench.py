"""Per-call times of the step with the source-image gradient at bench.py's config-3 finest level (B=12, S=2, 192x640) with
bench.py's random per-pixel depth and with a smooth depth map: HIP events around each C call, 200 steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import depth_estimation_amd as de

dev = torch.device("cuda", 0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
for kind in (sys.argv[1:] or ["random", "smooth"]):
    sc = bench.Scale(de, 12, 2, 192, 640, dev, 0, True)
    if kind == "smooth":
        sc.depth.copy_(torch.linspace(80, 5, 192, device=dev).view(1, 1, 192, 1).expand_as(sc.depth) + 0.5 * torch.rand_like(sc.depth))
    s = st.cuda_stream
    for _ in range(300):
        sc.step(s)
    torch.cuda.synchronize()
    ev = {n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)] for n in sc.names}
    for i in range(200):
        for n in sc.names:
            ev[n][i][0].record(); sc.calls[n](s); ev[n][i][1].record()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(200):
        sc.step(s)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / 200 * 1e3
    print(kind, {n: round(sum(a.elapsed_time(b) for a, b in ev[n]) / 200, 4) for n in sc.names}, "one call", round(one, 4), "ms")
