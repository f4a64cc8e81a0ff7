"""ds_warp_bwd with the image gradient: random per-pixel depth (bench.py's synthetic batch: a few near pixels among far
ones in every wave row) against a smooth depth map (what a network predicts), same poses."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import depth_estimation_amd as de
dev = torch.device("cuda", 0)
for B, S, H, W in ((12, 2, 192, 640), (64, 2, 320, 1024)):
    for kind in ("random", "smooth", "none"):
        sc = bench.Scale(de, B, S, H, W, dev, 0, kind != "none")
        if kind == "smooth":
            yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
            sc.depth.copy_((4 + 60 * (1 - yy / H) + 6 * torch.sin(xx / 40.0)).expand(B, S, H, W))
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            sc.step(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        tot = 0.0
        for _ in range(n):
            if sc.g_src is not None:
                sc.g_src.zero_()
            e0.record()
            sc.calls["warp_bwd"](st)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        print(f"B={B} {H}x{W} depth={kind:7s} warp_bwd {tot / n * 1e3:8.1f} us")
