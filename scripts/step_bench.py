"""Wall time of the whole loss step through the Python layer (fused.train_step_loss + backward: warp, photometric loss,
auto-mask ratio, smoothness and their autograd), as a training loop would call it.  Compare with bench.py, which times
the C-ABI step alone."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import depth_estimation_amd as de
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for B, S, H, W in ((12, 2, 192, 640), (64, 2, 320, 1024)):
    src = torch.rand(B, S, 3, H, W, generator=g).to(dev)
    tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
    depth = (1 + 79 * torch.rand(B, S, 1, H, W, generator=g)).to(dev).requires_grad_(True)
    vec = (0.01 * torch.randn(B * S, 6, generator=g)).to(dev).requires_grad_(True)
    K, Kinv = [t.to(dev) for t in de.fused.intrinsics(W, H)]

    for eager in (False, True):   # two C calls each way / gradients formed in the forward's one call
        def step():
            depth.grad = None
            vec.grad = None
            pose = de.synthesis.construct_pose_batch(vec, dev).reshape(B, S, 3, 4)
            r = de.fused.train_step_loss(src, tgt, depth, pose, K, Kinv, eager_grads=eager)
            r["total"].backward()

        for _ in range(10):
            step()
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"train_step_loss fwd+bwd B={B} S={S} {H}x{W} eager_grads={eager}: {ms:.3f} ms per step, "
              f"{B*S*H*W/ms/1e6:.2f} Gpix/s warped")
