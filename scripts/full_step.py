"""BASELINE config 5: the whole train.py step, batch-sharded over the ranks, with the gradient all-reduce.

    python bench.py --workload full_step [--gpus N] [--steps K] [--warmup W]

What a step does (reference: train.py:39-109; optimizer set train.py:140; frozen encoder depth_encoder.py:28):

    pose vectors   = PoseNet(target x S, sources)                      train.py:45     stand-in network
    pose matrices  = synthesis.construct_pose_batch                    train.py:46     HIP (ds_pose_from_vec_*)
    candidates     = plane sweep -> cost volume -> soft arg-min        train.py:49-60  HIP (fused.candidate_depths), no grad
    depth          = ConvGRU(Decoder(candidate, target)) * MAX_DEPTH   train.py:63-69  stand-in networks
    loss           = sum over samples of the auto-masked reprojection  train.py:72-89  HIP (fused.train_step_loss)
                     loss + 0.02 * smoothness
    backward, gradient all-reduce, 3 x clip_grad_norm_(1.0), Adam      train.py:102-109

The networks are out of scope (dense convolutions, and PoseNet / the encoder need torchvision weights
that are not available offline), so they are STAND-INS: small conv stacks with exactly the reference's trainable parameter
counts -- pose 11,182,662, decoder 25,908,589, ConvGRU 491,577 = 37,582,828 fp32 = 150.3 MB of gradients per step -- so
that the collective, the clipping and the optimizer move what the reference's would.  The frozen encoder has no gradient
and takes no part in any of that; it is left out.

Every rank owns B samples (weak scaling).  The samples of a batch are SUMMED into the loss (train.py:88), so the ranks'
gradients are summed too (GradBuckets(average=False)): N ranks x B samples give the gradient of one process with N*B.
The all-reduce is launched per 25 MB bucket from a gradient hook as soon as backward has filled the bucket, and overlaps
with the rest of backward (dist.GradBuckets.attach).
"""
from __future__ import annotations

import json
import os
import sys
import time
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# trainable parameters of the reference's three optimised networks (counted from networks/*.py)
REF_PARAMS = dict(pose=11_182_662, decoder=25_908_589, convgru=491_577)
MAX_DEPTH = 80.0       # constants.py
COST_H, COST_W = 96, 320


class _StandIn(nn.Module):
    """A small conv body plus a flat filler parameter that brings the module to exactly `n_params` trainable parameters.
    The filler enters the output with weight zero, so it gets a (zero) gradient every step: it is all-reduced, clipped and
    stepped by Adam like the real weights it stands for."""

    def __init__(self, n_params: int):
        super().__init__()
        self._target = n_params

    def _fill(self):
        body = sum(p.numel() for p in self.parameters())
        if body > self._target:
            raise ValueError(f"stand-in body has {body} parameters, more than the {self._target} it stands for")
        self.filler = nn.Parameter(torch.zeros(self._target - body))

    def _touch(self, x):
        return x + self.filler.sum() * 0.0


class PoseStandIn(_StandIn):
    """(target, source) image pairs (N,3,H,W) x2 -> (N,6) small pose vectors (networks/posenet.py:22-29)."""

    def __init__(self, n_params=REF_PARAMS["pose"], width=16):
        super().__init__(n_params)
        self.c1 = nn.Conv2d(6, width, 7, stride=4, padding=3)
        self.c2 = nn.Conv2d(width, 2 * width, 3, stride=2, padding=1)
        self.fc = nn.Linear(2 * width * 4 * 8, 6)
        self._fill()

    def forward(self, target, src):
        x = F.relu(self.c1(torch.cat([target, src], dim=1)))
        x = F.relu(self.c2(x))
        x = F.adaptive_avg_pool2d(x, (4, 8)).flatten(1)
        return self._touch(0.01 * self.fc(x))


class DecoderStandIn(_StandIn):
    """candidate depth (N,1,H,W) + target (N,3,H,W) -> (N,1,H,W) (networks/depth_decoder.py:34-50)."""

    def __init__(self, n_params=REF_PARAMS["decoder"], width=8):
        super().__init__(n_params)
        self.c1 = nn.Conv2d(4, width, 3, padding=1)
        self.c2 = nn.Conv2d(width, 1, 3, padding=1)
        self._fill()

    def forward(self, cand, target):
        x = F.relu(self.c1(torch.cat([cand, target], dim=1)))
        return self._touch(self.c2(x))


class ConvGruStandIn(_StandIn):
    """(N,1,H,W) -> (N,1,H,W) in (0,1) (networks/depth_convgru.py:31-50; the persistent state is left out)."""

    def __init__(self, n_params=REF_PARAMS["convgru"], width=4):
        super().__init__(n_params)
        self.c1 = nn.Conv2d(1, width, 3, padding=1)
        self.c2 = nn.Conv2d(width, 1, 3, padding=1)
        self._fill()

    def forward(self, x):
        return self._touch(torch.sigmoid(self.c2(torch.tanh(self.c1(x)))))


def build_networks(device, counts=None, seed=0):
    counts = counts or REF_PARAMS
    torch.manual_seed(seed)   # the same initial weights on every rank (replicas)
    nets = SimpleNamespace(pose=PoseStandIn(counts["pose"]), decoder=DecoderStandIn(counts["decoder"]),
                           convgru=ConvGruStandIn(counts["convgru"]))
    for m in (nets.pose, nets.decoder, nets.convgru):
        m.to(device)
    return nets


def trainable(nets):
    return [p for m in (nets.pose, nets.decoder, nets.convgru) for p in m.parameters()]


def train_step(nets, opt, buckets, imgs, path, max_depth=MAX_DEPTH):
    """One optimisation step on this rank's samples imgs (B, S+1, 3, H, W); the last frame of a sample is the target
    (train.py:42).  `path` supplies the hot-path functions (HIP in bench.py, torch stand-ins in the CPU test):
    construct_pose_batch(vec) -> (N,3,4); candidate_depths(src, target, pose) -> (B,S,1,H,W); loss(src, target, depth, pose)
    -> scalar summed over the samples.  Returns the loss tensor (no host sync)."""
    B, F_, C, H, W = imgs.shape
    S = F_ - 1
    target, src = imgs[:, -1].contiguous(), imgs[:, :-1].contiguous()
    tgt_rep = target[:, None].expand(B, S, C, H, W).reshape(B * S, C, H, W)
    vec = nets.pose(tgt_rep, src.reshape(B * S, C, H, W))                       # train.py:45
    pose = path.construct_pose_batch(vec).reshape(B, S, 3, 4)                   # train.py:46
    with torch.no_grad():
        cand = path.candidate_depths(src, target, pose.detach())                # train.py:49-60
    d = nets.convgru(nets.decoder(cand.reshape(B * S, 1, H, W), tgt_rep))       # train.py:63-69
    depth = (d * max_depth).reshape(B, S, 1, H, W)
    loss = path.loss(src, target, depth, pose)                                  # train.py:72-89
    opt.zero_grad(set_to_none=False)                                            # train.py:102
    if buckets is not None:
        buckets.begin()
    loss.backward()                                                             # train.py:103 (+ hooks: bucket all-reduces)
    if buckets is not None:
        buckets.finish()
    for m in (nets.pose, nets.decoder, nets.convgru):                           # train.py:106-108
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()                                                                  # train.py:109
    return loss.detach()


def hip_path(de, H, W, device):
    """The hot-path functions on the GPU: depth_estimation_amd's fused entry points (no CPU fallback)."""
    import numpy as np
    K, Kinv = de.fused.intrinsics(W, H, device)
    Kc, Kcinv = de.fused.intrinsics(COST_W, COST_H, device)                     # constants.py:33-37
    depths = np.array([np.exp((x + 1) * np.log(2) / 80) - 1 for x in range(80)]) * MAX_DEPTH   # SID planes, constants.py:19
    return SimpleNamespace(
        construct_pose_batch=lambda vec: de.synthesis.construct_pose_batch(vec, device),
        candidate_depths=lambda src, tgt, pose: de.fused.candidate_depths(src, tgt, pose, Kc, Kcinv, depths, (COST_H, COST_W)),
        loss=lambda src, tgt, depth, pose: de.fused.train_step_loss(src, tgt, depth, pose, K, Kinv, eager_grads=True)["total"])


def torch_stand_in_path():
    """Differentiable torch stand-ins for the three hot-path functions, shapes as the HIP ones.  NOT the product and not
    its arithmetic: they exist so that the launcher's --dry run and the CPU tests (gloo, no GPU) can drive this harness --
    networks, buckets, clipping, Adam -- end to end.  The product path is hip_path() and has no CPU form."""
    def construct_pose_batch(vec):
        n = vec.shape[0]
        eye = torch.eye(3, device=vec.device).expand(n, 3, 3)
        return torch.cat([eye + 0.1 * vec[:, :3, None] * vec[:, None, :3], vec[:, 3:, None]], dim=2)

    def candidate_depths(src, target, pose):
        return 0.1 + 0.8 * (src.mean(2, keepdim=True) - target[:, None].mean(2, keepdim=True)).abs()

    def loss(src, target, depth, pose):
        photo = ((src - target[:, None]).abs().mean(2, keepdim=True) * depth / 80.0).mean(dim=(1, 2, 3, 4))
        return photo.sum() + 0.01 * pose.pow(2).mean(dim=(1, 2, 3)).sum()   # summed over the samples (train.py:88)

    return SimpleNamespace(construct_pose_batch=construct_pose_batch, candidate_depths=candidate_depths, loss=loss)


def main(args, rank, local_rank, world, dry=False):
    """dry: the same harness on the CPU over gloo -- launcher, rendezvous, bucketed all-reduce from the backward hooks,
    clipping, Adam -- with torch stand-ins for the hot path, tiny networks and images; no kernels, `value` 0."""
    import torch.distributed as dist
    if dry:
        dev = torch.device("cpu")
        torch.set_num_threads(1)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from depth_estimation_amd.dist import GradBuckets
        B, S, H, W = 1, 2, 16, 24
        counts, bucket_bytes = dict(pose=40_000, decoder=3_001, convgru=777), 16 << 10
        path = torch_stand_in_path()
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py --workload full_step needs a GPU (the product path has no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        import depth_estimation_amd as de
        from depth_estimation_amd.dist import GradBuckets
        B, S, H, W = 2, 3, 192, 640                                              # constants.py:6-7 (BATCH_SIZE, SEQ_LEN - 1)
        counts, bucket_bytes = None, 25 << 20
        path = hip_path(de, H, W, dev)
    nets = build_networks(dev, counts)
    params = trainable(nets)
    n_params = sum(p.numel() for p in params)
    opt = torch.optim.Adam(params)                                               # train.py:140
    buckets = GradBuckets(params, bucket_bytes=bucket_bytes, average=False)
    buckets.attach()
    g = torch.Generator().manual_seed(100 + rank)
    imgs = torch.rand(B, S + 1, 3, H, W, generator=g).to(dev)

    def barrier():
        if not dry:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def run(k, bk):
        for _ in range(k):
            train_step(nets, opt, bk, imgs, path)

    run(args.warmup, buckets)
    barrier()
    t0 = time.perf_counter()
    run(args.steps, buckets)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # replicas must still hold the same weights after the steps (same initial weights, summed gradients, same Adam)
    digest = torch.stack([p.detach().double().sum() for p in params]).sum().reshape(1).to(dev)
    spread = 0.0
    if world > 1:
        lo, hi = digest.clone(), digest.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = float((hi - lo).item())
    # the same steps without the collective (what the all-reduce adds when overlapped; the replicas drift apart from here
    # on, which is why the spread above is taken first), and the collective alone
    barrier()
    t0 = time.perf_counter()
    run(args.steps, None)
    barrier()
    no_comm = time.perf_counter() - t0
    comm_ms = 0.0
    if world > 1:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            buckets.all_reduce()
        barrier()
        comm_ms = (time.perf_counter() - t0) / args.steps * 1e3
    # what the process group itself saw: a SUM of ones over the ranks must read N
    ranks_summed = 1
    if world > 1:
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        ranks_summed = int(ones.item())
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": "warped px/s (Gpix/s) fwd+bwd reproj loss @192x640, % HBM roofline, 1-8 GPU",
            "value": 0.0 if dry else round(world * B * S * H * W * args.steps / elapsed / 1e9, 5), "unit": "Gpix/s",
            "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "full_step", "samples_per_gpu": B, "sources_per_target": S, "height": H, "width": W,
                       "networks": "stand-ins with the reference's trainable parameter counts", "params": n_params,
                       "parallelism": f"batch-sharded x{world}, gradient all-reduce (RCCL), SUM over ranks"},
            "samples_per_s": round(world * B * args.steps / elapsed, 2),
            "world_size": dist.get_world_size() if world > 1 else 1, "ranks_summed": ranks_summed,
            "backend": dist.get_backend() if world > 1 else None, "replica_weight_spread": spread,
            "grad_bytes_all_reduced": buckets.nbytes(), "buckets": len(buckets.buckets), "bucket_bytes": bucket_bytes,
            "comm_ms": round(comm_ms, 4), "comm_exposed_ms": round(ms - no_comm / args.steps * 1e3, 4),
            "ms_per_step_without_collective": round(no_comm / args.steps * 1e3, 4),
        }
        if dry:
            line["dry"] = True
            line["config"]["networks"] = "tiny stand-ins; torch stand-ins for the hot path (launcher / collective check only)"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
