import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import depth_estimation_amd as de
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
B, S, H, W = 12, 2, 192, 640
src = torch.rand(B, S, 3, H, W, generator=g).to(dev)
tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
depth = (1 + 79 * torch.rand(B, S, 1, H, W, generator=g)).to(dev).requires_grad_(True)
vec = (0.01 * torch.randn(B * S, 6, generator=g)).to(dev).requires_grad_(True)
K, Kinv = [t.to(dev) for t in de.fused.intrinsics(W, H)]
def step():
    depth.grad = None; vec.grad = None
    pose = de.synthesis.construct_pose_batch(vec, dev).reshape(B, S, 3, 4)
    r = de.fused.train_step_loss(src, tgt, depth, pose, K, Kinv, eager_grads=True)
    r["total"].backward()
for _ in range(30): step()
torch.cuda.synchronize()
