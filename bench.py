#!/usr/bin/env python3
"""bench.py -- fused reprojection-loss forward+backward on N MI355X GPUs (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--image-grad] [--no-cpu]

A "step" is one pass of the hot path (train.py:72-92 batched: warp fwd -> SSIM+L1/min/auto-mask fwd ->
its backward -> warp backward with depth and pose gradients) over one synthetic batch that is already
resident in HBM.  Kernels are called through the C ABI (include/depthsynth.h) with pre-allocated outputs, the
whole batch on ONE stream, every step ordered after the previous one (train.py:102-109 joins every batch).
The batch is sharded over ranks with no data-path collective (weak scaling: every rank owns B targets);
the only cross-rank traffic is the barrier and a MAX all-reduce of the elapsed time.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts its own N ranks (fresh child processes, one per
GPU, RCCL rendezvous on 127.0.0.1) and waits for them; under torch.distributed.run it uses the ranks it is given.
`--workload full_step` is BASELINE config 5: the whole train.py step with stand-in networks and the gradient
all-reduce (scripts/full_step.py holds the harness).

Rank 0 prints ONE JSON line.  `value` = warped pixels (B*S*H*W per rank, all ranks) per second.
`roofline` is for the dominant kernel (HIP events on the launch stream, inside this run); `roofline.valu` prices the
same kernel against the VALU issue peak (instruction counts from the committed PMC pass, profiles/pmc.json).
`cpu_baseline` = the CPU oracle (a C port of the reference's arithmetic) on a bounded sample, on the host's cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json metric: "fwd+bwd reproj loss @192x640" on the config-2/3 batch
    "b12_s2_192x640_fwdbwd": dict(B=12, S=2, H=192, W=640, scales=1),
    # config 1: "fused forward warp+loss" -- the forward half alone (ds_loss_step_fwd)
    "b12_s2_192x640_fwd": dict(B=12, S=2, H=192, W=640, scales=1, fwd_only=True),
    # the reference's own sequence length (constants.py: SEQ_LEN = 4, three sources per target)
    "b8_s3_192x640_fwdbwd": dict(B=8, S=3, H=192, W=640, scales=1),
    # config 3: the same over a 4-scale pyramid (each scale an independent call, K rows scaled per scale)
    # (the backward also scatters the source-image gradient)
    "b12_s2_192x640_pyramid4_fwdbwd": dict(B=12, S=2, H=192, W=640, scales=4, image_grad=True),
    # config 4: the HBM-resident-set-exceeds-cache case; the step ends with the depth-metrics reduction on (B,1,H,W)
    "b64_s2_320x1024_fwdbwd": dict(B=64, S=2, H=320, W=1024, scales=1, metrics=True),
    # train.py:49-60 for a batch (not in the headline): target -> 96x320 (bilinear), 80-plane sweep of the
    # 192x640 sources with the fused cost volume + soft arg-min, candidate depth -> 192x640 (bicubic); forward only
    "b12_s2_sweep80_96x320": dict(B=12, S=2, H=192, W=640, scales=1, sweep=dict(D=80, h=96, w=320)),
    # config 5: the whole train.py step (stand-in networks, fused loss path, gradient all-reduce, clip, Adam)
    "full_step": dict(B=2, S=3, H=192, W=640, scales=1),
}
# what the default line also reports (round-2 verdict item 2): every other config, joined, one stream, <= ~1 s each
ALSO = ("b12_s2_192x640_fwd", "b8_s3_192x640_fwdbwd", "b12_s2_192x640_pyramid4_fwdbwd", "b64_s2_320x1024_fwdbwd",
        "b12_s2_sweep80_96x320")
METRIC = "warped px/s (Gpix/s) fwd+bwd reproj loss @192x640, % HBM roofline, 1-8 GPU"
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
SPIN_UP_S = 0.3              # untimed steps before the warm-up: the device leaves its idle power state (see main())
VALU_ISSUE_NS = 0.83         # one fp32 VALU wave instruction per SIMD per 2 cycles at 2.4 GHz (profiles/r01_ubench4_valu_stream.txt)


def step_alg_bytes_per_target_px(S, C=3, image_grad=False):
    """forward 4(S+2CS+C+2) + backward 4(2S+2CS+C+2) [+4CS with the image gradient] bytes per TARGET pixel
    (= 80 B per warped pixel at S=2, C=3; 92 with the image gradient)."""
    return 4 * (S + 2 * C * S + C + 2) + 4 * (2 * S + 2 * C * S + C + 2) + (4 * C * S if image_grad else 0)


# pose vectors of the synthetic batch are N(0, POSE_SIGMA^2) per component; the environment variable is for traffic
# diagnostics only (0 = identity warp: every gather lands on the pixel's own texel) and shows up in config.pose_sigma
POSE_SIGMA = float(os.environ.get("DS_BENCH_POSE_SIGMA", "0.01"))


class Scale:
    """All device buffers of one (B,S,H,W) problem + the C-ABI calls of a step."""

    def __init__(self, de, B, S, H, W, dev, seed, image_grad, metrics=False, fwd_only=False):
        from depth_estimation_amd import _lib as L
        self.L, self.lib = L, L.lib()
        self.B, self.S, self.H, self.W, self.N = B, S, H, W, B * S
        g = torch.Generator().manual_seed(seed)
        f = dict(dtype=torch.float32, device=dev)
        self.src = torch.rand(B, S, 3, H, W, generator=g).to(dev)
        self.tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
        self.depth = (1 + 79 * torch.rand(B, S, H, W, generator=g)).to(dev)
        vec = (POSE_SIGMA * torch.randn(B * S, 6, generator=g)).to(dev)
        self.pose = de.ops.pose_from_vec(vec).contiguous()
        K, Kinv = de.fused.intrinsics(W, H)
        self.K, self.Kinv = K.to(dev), Kinv.to(dev)
        self.warped = torch.empty(B, S, 3, H, W, **f)
        self.out = torch.empty(B, H, W, **f)
        self.idt = torch.empty(B, H, W, **f)
        self.aux = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
        self.sums = torch.empty(B, 2, **f)
        self.g_sums = torch.ones(B, **f)
        self.g_warped = torch.empty(B, S, 3, H, W, **f)
        self.g_depth = torch.empty(B, S, H, W, **f)
        self.g_pose = torch.empty(B, S, 3, 4, **f)
        self.g_src = torch.empty(B, S, 3, H, W, **f) if image_grad else None   # written whole by the call (no clearing)
        self.ws_loss_n = self.lib.ds_fused_loss_workspace_bytes(B, H, W)
        self.ws_warp_n = (self.lib.ds_warp_bwd_image_workspace_bytes(self.N, 3, H, W, H, W) if image_grad
                          else self.lib.ds_warp_bwd_workspace_bytes(self.N, H, W))
        self.ws_loss = torch.empty(self.ws_loss_n, dtype=torch.uint8, device=dev)
        self.ws_warp = torch.empty(self.ws_warp_n, dtype=torch.uint8, device=dev)
        p = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
        lib, N = self.lib, self.N
        # The launches of a step, individually callable (per-kernel timing).  Without the image gradient and with S <= 2
        # the step is what ds_loss_step_fwdbwd launches: warp + identity map in one launch, then forward+backward of the
        # warped groups with the warp's autograd in one kernel; otherwise the four classic calls.
        self.fused_path = S <= 2 or (S == 3 and not image_grad)
        if metrics:
            # test.py:73-81 on the first source's depth map; ground truth U(1,80) with 30 % invalid (zero) pixels
            self.gt = (1 + 79 * torch.rand(B, 1, H, W, generator=g)).to(dev)
            self.gt[torch.rand(B, 1, H, W, generator=g).to(dev) < 0.3] = 0.0
            self.pred = self.depth[:, 0].contiguous()
            self.met = torch.empty(8, **f)
            self.ws_met_n = self.lib.ds_metrics_workspace_bytes(B, H, W)
            self.ws_met = torch.empty(self.ws_met_n, dtype=torch.uint8, device=dev)
        self.ws_fb_n = self.lib.ds_fused_loss_fb_warp_bwd_workspace_bytes(B, S, H, W)
        self.ws_fb = torch.empty(max(self.ws_fb_n, 16), dtype=torch.uint8, device=dev)
        px, tp = self.N * H * W, B * H * W
        self.units = px                                          # warped pixels per step
        warp_bwd = lambda st: lib.ds_warp_bwd(p(self.K), p(self.Kinv), p(self.pose), p(self.depth), p(self.src),  # noqa: E731
                                              p(self.g_warped), N, 3, H, W, H, W, p(self.g_depth), p(self.g_pose),
                                              p(self.g_src), p(self.ws_warp), self.ws_warp_n, st)
        # algorithmic HBM bytes of each call: every input read once, every output written once
        warp_identity = lambda st: lib.ds_warp_identity_fwd(p(self.K), p(self.Kinv), p(self.pose), p(self.depth),  # noqa: E731
                                                            p(self.src), p(self.tgt), B, S, H, W, p(self.warped), p(self.idt), st)
        if self.fused_path and image_grad:
            # with the image gradient d loss / d warped goes through memory to the scattering warp backward kernel
            self.ws_fb2_n = self.lib.ds_fused_loss_fb_workspace_bytes(B, H, W)
            self.ws_fb2 = torch.empty(max(self.ws_fb2_n, 16), dtype=torch.uint8, device=dev)
            self.calls = {
                "warp_identity": warp_identity,
                "loss_fb": lambda st: lib.ds_fused_loss_fb(p(self.warped), p(self.tgt), p(self.idt), 0, B, S, H, W, p(self.out),
                                                           p(self.aux), p(self.sums), p(self.g_warped), p(self.ws_fb2),
                                                           self.ws_fb2_n, st),
                "warp_bwd": warp_bwd,
            }
            self.alg_bytes = {
                "warp_identity": px * (4 + 12 + 12) + tp * (12 + 4),
                "loss_fb": px * (12 + 12) + tp * (12 + 4 + 4 + 1),     # warped + target + idt -> g_warped, out, aux
                "warp_bwd": px * (4 + 12 + 12 + 4 + 12),               # depth + src + g_warped -> g_depth, g_src
            }
        elif self.fused_path:
            self.calls = {
                "warp_identity": lambda st: lib.ds_warp_identity_fwd(p(self.K), p(self.Kinv), p(self.pose), p(self.depth),
                                                                     p(self.src), p(self.tgt), B, S, H, W, p(self.warped),
                                                                     p(self.idt), st),
                "loss_fb_warp_bwd": lambda st: lib.ds_fused_loss_fb_warp_bwd(
                    p(self.K), p(self.Kinv), p(self.pose), p(self.depth), p(self.src), p(self.warped), p(self.tgt), p(self.idt), 0,
                    B, S, H, W, p(self.out), p(self.aux), p(self.sums), 0, p(self.g_depth), p(self.g_pose), p(self.ws_fb),
                    self.ws_fb_n, st),
            }
            self.alg_bytes = {
                "warp_identity": px * (4 + 12 + 12) + tp * (12 + 4),   # depth + src -> warped; + target -> idt
                # warped + depth + src (gather) -> g_depth; + target + idt -> out, aux (d loss / d warped stays on chip)
                "loss_fb_warp_bwd": px * (12 + 4 + 12 + 4) + tp * (12 + 4 + 4 + 1),
            }
        else:
            self.calls = {
                "warp_fwd": lambda st: lib.ds_warp_fwd(p(self.K), p(self.Kinv), p(self.pose), p(self.depth), p(self.src), N, 3,
                                                        H, W, H, W, p(self.warped), st),
                "loss_fwd": lambda st: lib.ds_fused_loss_fwd(p(self.warped), p(self.src), p(self.tgt), B, S, H, W, p(self.out),
                                                             p(self.idt), p(self.aux), p(self.sums), p(self.ws_loss),
                                                             self.ws_loss_n, st),
                "loss_bwd": lambda st: lib.ds_fused_loss_bwd(p(self.warped), p(self.tgt), p(self.idt), p(self.aux),
                                                             p(self.g_sums), B, S, H, W, p(self.g_warped), st),
                "warp_bwd": warp_bwd,
            }
            self.alg_bytes = {
                "warp_fwd": px * (4 + 12 + 12),                       # depth + src (gather) -> warped
                "loss_fwd": px * (12 + 12) + tp * (12 + 4 + 4 + 1),   # warped + src + target -> out, idt, aux
                "loss_bwd": px * (12 + 12) + tp * (12 + 4 + 1),       # warped + target + idt + aux -> g_warped
                "warp_bwd": px * (4 + 12 + 12 + 4 + (12 if image_grad else 0)),  # depth + src + g_warped -> g_depth [+ g_src]
            }
        if fwd_only:
            self.ws_fwd_n = self.lib.ds_loss_step_workspace_bytes(B, S, H, W)
            self.ws_fwd = torch.empty(max(self.ws_fwd_n, 16), dtype=torch.uint8, device=dev)
            self.calls = {"loss_step_fwd": lambda st: lib.ds_loss_step_fwd(
                p(self.K), p(self.Kinv), p(self.pose), p(self.depth), p(self.src), p(self.tgt), B, S, H, W, p(self.warped),
                p(self.out), p(self.idt), p(self.aux), p(self.sums), p(self.ws_fwd), self.ws_fwd_n, st)}
            # SURVEY 8d forward: depth + src (gather) -> warped, warped + src + target -> out, idt, aux
            self.alg_bytes = {"loss_step_fwd": tp * 4 * (S + 2 * 3 * S + 3 + 2)}
        self.names = tuple(self.calls) + (("metrics",) if metrics else ())
        # the step as ONE call (forward + backward launches and a single merged finalize; with the image gradient also the
        # clearing of g_src)
        self.ws_step_n = (self.lib.ds_loss_step_fwdbwd_image_workspace_bytes(B, S, H, W) if image_grad
                          else self.lib.ds_loss_step_fwdbwd_workspace_bytes(B, S, H, W))
        self.ws_step = torch.empty(self.ws_step_n, dtype=torch.uint8, device=dev)
        self.one_call = lambda st: lib.ds_loss_step_fwdbwd(
            p(self.K), p(self.Kinv), p(self.pose), p(self.depth), p(self.src), p(self.tgt), 0, B, S, H, W, p(self.warped),
            p(self.out), p(self.idt), p(self.aux), p(self.sums), p(self.g_warped), p(self.g_depth), p(self.g_pose),
            p(self.g_src), p(self.ws_step), self.ws_step_n, st)
        if metrics:
            self.calls["metrics"] = lambda st: lib.ds_depth_metrics(p(self.pred), p(self.gt), B, H, W, p(self.met), 0,
                                                                    p(self.ws_met), self.ws_met_n, st)
            self.alg_bytes["metrics"] = tp * 8                    # pred + gt
        # what BASELINE.md prices the step with: SURVEY 8d's bytes of the fused path (no warped / g_warped round trips)
        self.step_bytes = tp * step_alg_bytes_per_target_px(S, 3, image_grad) + (tp * 8 if metrics else 0)
        if fwd_only:
            self.one_call = self.calls["loss_step_fwd"]
            self.step_bytes = self.alg_bytes["loss_step_fwd"]

    def step(self, st, separate=False):
        if self.one_call is not None and not separate:
            rc = self.one_call(st)
            if rc != 0:
                self.L.check(rc, "ds_loss_step_fwdbwd")
            if "metrics" in self.calls:
                rc = self.calls["metrics"](st)
                if rc != 0:
                    self.L.check(rc, "metrics")
            return
        for name in self.names:
            rc = self.calls[name](st)
            if rc != 0:
                self.L.check(rc, name)


class Sweep:
    """train.py:49-60 for B targets x S sources through the C ABI with pre-allocated outputs: ds_resize_bilinear ->
    ds_plane_sweep_depth_batch -> ds_resize_bicubic.  Unit of work: a plane pixel (one source warped onto one plane at one
    cost-grid pixel), B*S*D*h*w of them per step."""

    def __init__(self, de, B, S, H, W, D, h, w, dev, seed):
        import ctypes
        import numpy as np
        from depth_estimation_amd import _lib as L
        self.L, self.lib = L, L.lib()
        self.B, self.S, self.H, self.W, self.N = B, S, H, W, B * S
        self.D, self.h, self.w = D, h, w
        g = torch.Generator().manual_seed(seed)
        f = dict(dtype=torch.float32, device=dev)
        self.src = torch.rand(B, S, 3, H, W, generator=g).to(dev)
        self.tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
        vec = (0.01 * torch.randn(B * S, 6, generator=g)).to(dev)
        self.pose = de.ops.pose_from_vec(vec).contiguous()
        K, Kinv = de.fused.intrinsics(w, h)                       # constants.py:33-37: intrinsics of the COST grid
        self.K, self.Kinv = K.to(dev), Kinv.to(dev)
        self.depths = np.ascontiguousarray(
            np.array([np.exp((x + 1) * np.log(2) / D) - 1 for x in range(D)]) * 80.0, dtype=np.float64)   # constants.py:19
        self.dp = self.depths.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self.small = torch.empty(B, 3, h, w, **f)
        self.cand = torch.empty(B, S, 1, h, w, **f)
        self.full = torch.empty(B, S, 1, H, W, **f)
        p = lambda t: t.data_ptr()  # noqa: E731
        lib = self.lib
        self.calls = {
            "resize_bilinear": lambda st: lib.ds_resize_bilinear(p(self.tgt), B * 3, H, W, h, w, p(self.small), st),
            "plane_sweep": lambda st: lib.ds_plane_sweep_depth_batch(p(self.K), p(self.Kinv), p(self.pose), self.dp, D,
                                                                     p(self.src), p(self.small), B, S, 3, H, W, h, w, 1,
                                                                     p(self.cand), st),
            "resize_bicubic": lambda st: lib.ds_resize_bicubic(p(self.cand), B * S, h, w, H, W, p(self.full), st),
        }
        self.names = tuple(self.calls)
        # SURVEY 8d "Plane sweep": per source the sampled source REGION (the sweep normalises with the source dims, so a
        # 96x320-unit grid only reaches the top-left h x w texels, synthesis.py:49) + the small target read, depth written
        self.alg_bytes = {
            "resize_bilinear": B * 3 * (H * W + h * w) * 4,
            "plane_sweep": self.N * (3 * h * w + 3 * h * w + h * w) * 4,
            "resize_bicubic": self.N * (h * w + H * W) * 4,
        }
        self.step_bytes = sum(self.alg_bytes.values())
        self.units = self.N * D * h * w

    def step(self, st, separate=False):
        for name in self.names:
            rc = self.calls[name](st)
            if rc != 0:
                self.L.check(rc, name)


def build_problem(de, wl, dev, rank, image_grad, parts=1):
    """One Scale per (pyramid level, sub-batch).  With parts > 1 the B targets of a level are split into `parts`
    independent sub-batches (own buffers, own workspaces), which the pipelined variant launches on `parts` HIP streams."""
    if "sweep" in wl:
        sw = wl["sweep"]
        return [Sweep(de, wl["B"], wl["S"], wl["H"], wl["W"], sw["D"], sw["h"], sw["w"], dev, 1000 * rank)]
    scales = []
    for i in range(wl["scales"]):
        # small pyramid levels are launch-bound: splitting them only adds launches
        lvl_parts = parts if wl["B"] * (wl["H"] >> i) * (wl["W"] >> i) >= (1 << 20) else 1
        for j in range(lvl_parts):
            b = wl["B"] // lvl_parts + (1 if j < wl["B"] % lvl_parts else 0)
            if b > 0:
                scales.append(Scale(de, b, wl["S"], wl["H"] >> i, wl["W"] >> i, dev, 1000 * rank + 10 * i + j, image_grad,
                                    wl.get("metrics", False), wl.get("fwd_only", False)))
    return scales


def host_threads():
    """Threads for the CPU leg: the box exports its CPU share as OMP_NUM_THREADS (os.cpu_count() shows the whole machine)."""
    try:
        n = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        n = 0
    return n if n > 0 else (os.cpu_count() or 1)


def cpu_baseline(wl, budget_s=8.0):
    """The CPU oracle (C port) on whole samples of the same workload: ~budget_s seconds on all host threads with OpenMP over
    image rows inside a sample (per-sample loop as train.py:39), ~budget_s / 2 on one thread, ~budget_s with one sample per
    thread; the better of the two all-core figures is `value`."""
    import numpy as np
    from oracle import ds_oracle as o
    o.build()
    S, H, W = wl["S"], wl["H"], wl["W"]
    rng = np.random.default_rng(0)
    src = rng.random((S, 3, H, W), dtype=np.float32)
    tgt = rng.random((3, H, W), dtype=np.float32)
    depth = (1 + 79 * rng.random((S, H, W), dtype=np.float32)).astype(np.float32)
    pose = o.pose_from_vec((0.01 * rng.standard_normal((S, 6))).astype(np.float32))
    from depth_estimation_amd.fused import intrinsics
    K, Kinv = [a.numpy() for a in intrinsics(W, H)]

    if "sweep" in wl:
        sw = wl["sweep"]
        Kc, Kcinv = [a.numpy() for a in intrinsics(sw["w"], sw["h"])]
        planes = np.array([np.exp((x + 1) * np.log(2) / sw["D"]) - 1 for x in range(sw["D"])]) * 80.0

        def one_source():          # train.py:49-60 for one source of one target
            small = o.resize_bilinear(tgt, (sw["h"], sw["w"]))
            syn = o.synthesize_from_depths(Kc, Kcinv, pose[0], planes, sw["w"], sw["h"], src[0])
            cand = o.depth_from_costs(o.cost_volume(small, syn)[None])
            o.resize_bicubic(cand, (H, W))

        cores = host_threads()
        o.set_threads(cores)
        one_source()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s:
            one_source()
            n += 1
        dt = time.perf_counter() - t0
        return dict(value=n * sw["D"] * sw["h"] * sw["w"] / dt / 1e9, unit="G plane-px/s", cores=cores, kind="port",
                    sample=f"{n} sources x {sw['D']} planes at {sw['h']}x{sw['w']} from {H}x{W} (resize, sweep, cost volume, soft "
                           f"arg-min, resize) in {dt:.1f} s on {cores} threads, oracle/ds_oracle.c; os.cpu_count() = {os.cpu_count()}")

    def one_sample():
        warped = np.stack([o.warp(K, Kinv, pose[s], depth[s], src[s], H, W) for s in range(S)])
        la = o.loss_assembly(warped, src, tgt)
        gw = o.loss_assembly_bwd(warped, tgt, la["idt_loss"], la["argmin"])
        for s in range(S):
            o.warp_bwd(K, Kinv, pose[s], depth[s], src[s], gw[s], H, W, False)

    def timed(threads, budget):
        o.set_threads(threads)
        one_sample()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            one_sample()
            n += 1
        dt = time.perf_counter() - t0
        return n, dt, n * S * H * W / dt / 1e9

    def timed_sample_parallel(threads, budget):
        """`threads` host threads, each looping whole samples with a one-thread oracle (ctypes drops the GIL in the calls)."""
        import threading
        counts, stop = [0] * threads, time.perf_counter() + budget

        def work(i):
            o.set_threads(1)          # OpenMP's thread count is per calling thread
            while time.perf_counter() < stop:
                one_sample()
                counts[i] += 1

        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        return sum(counts), dt, sum(counts) * S * H * W / dt / 1e9

    cores = host_threads()
    n, dt, v = timed(cores, budget_s)
    n1, dt1, v1 = timed(1, budget_s / 2)
    ns, dts, vs = timed_sample_parallel(cores, budget_s)
    how = "OpenMP over rows, per-sample loop as train.py:39"
    line = dict(value=v, unit="Gpix/s", cores=cores, kind="port", one_thread_value=v1, rows_parallel_value=v,
                samples_parallel_value=vs)
    if vs > v:   # the better use of the same cores is the reported baseline
        line["value"], how = vs, "one sample per thread, one-thread oracle each"
    line["sample"] = (f"{ns if vs > v else n} samples (1 target x {S} sources, {H}x{W}, single scale) fwd+bwd in "
                      f"{dts if vs > v else dt:.1f} s on {cores} threads ({how}), oracle/ds_oracle.c; the other split of the cores "
                      f"({'OpenMP over rows inside a sample' if vs > v else 'one sample per thread'}): "
                      f"{n if vs > v else ns} samples in {dt if vs > v else dts:.1f} s; {n1} samples in {dt1:.1f} s on 1 thread; "
                      f"os.cpu_count() = {os.cpu_count()}")
    return line


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """Start n ranks of this script as fresh child processes (nothing in this process has touched the GPU) and wait."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def dry_run(args, rank, world):
    """--dry: the launcher / rendezvous / reduction path of a multi-rank run on the CPU (gloo), no kernels."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        size = dist.get_world_size()
    else:
        size = 1
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "Gpix/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry": True, "backend": "gloo", "world_size": size,
                          "max_elapsed_s": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def make_step(scales, main_stream, dev, fork):
    """The step of a workload on `main_stream`.  fork: a pyramid's levels are independent problems and the coarse ones
    cannot fill the chip (24x80 x 12 targets: 26 us for three launches that are all latency), so inside every step the levels
    below the finest are forked onto side streams and joined again before the step ends: the next step still starts only
    after everything of this one has finished."""
    st = main_stream.cuda_stream
    if not fork or len(scales) < 2:
        def step_one_stream():
            for sc in scales:
                sc.step(st)
        return step_one_stream
    side = [torch.cuda.Stream(device=dev) for _ in scales[1:]]
    ev_start = torch.cuda.Event()
    ev_done = [torch.cuda.Event() for _ in side]

    def step_forked():
        ev_start.record(main_stream)
        for sc, s_, e_ in zip(scales[1:], side, ev_done):
            s_.wait_event(ev_start)
            sc.step(s_.cuda_stream)
            e_.record(s_)
        scales[0].step(st)
        for e_ in ev_done:
            main_stream.wait_event(e_)
    return step_forked


def quick_line(de, name, dev, main_stream, budget_s=0.8):
    """One of the other workloads, joined, on the bench's stream: ~0.15 s of untimed steps, then as many timed steps as fit
    budget_s.  For the default line's `also` object; the workload's own full line comes from --workload NAME."""
    wl = WORKLOADS[name]
    scales = build_problem(de, wl, dev, 0, wl.get("image_grad", False), 1)
    res = {}
    for tag, fork in (("one_stream", False),) + ((("forked_levels", True),) if wl["scales"] > 1 else ()):
        step = make_step(scales, main_stream, dev, fork)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:
            for _ in range(5):
                step()
            torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        per = (time.perf_counter() - c0) / 5
        n = max(5, min(5000, int(budget_s / max(per, 1e-6))))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        units, nbytes = sum(sc.units for sc in scales), sum(sc.step_bytes for sc in scales)
        res[tag] = {"value": round(units / dt / 1e9, 4), "ms_per_step": round(dt * 1e3, 5), "steps": n,
                    "step_alg_bytes": nbytes, "step_frac_of_hbm_peak": round(nbytes / dt / 1e9 / HBM_PEAK_GBS, 4)}
    out = res["one_stream"]
    out["unit"] = "G plane-px/s" if "sweep" in wl else "Gpix/s"
    out["launch"] = "one stream, every step ordered after the previous one"
    if "forked_levels" in res:
        out["forked_levels"] = res["forked_levels"]
    del scales
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="b12_s2_192x640_fwdbwd", choices=sorted(WORKLOADS))
    ap.add_argument("--image-grad", action="store_true", help="also scatter the source-image gradient")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cold-steps", type=int, default=20, help="single steps timed after a cache flush (0 = skip)")
    ap.add_argument("--pipelined", type=int, default=3,
                    help="also report the un-joined variant: this many independent sub-batches, each on its own HIP "
                         "stream, never joined between steps (0 = skip); never the headline")
    ap.add_argument("--spin-up", type=float, default=SPIN_UP_S,
                    help="seconds of untimed steps before the warm-up (device out of its idle power state); 0 for profiler passes")
    ap.add_argument("--levels-one-stream", action="store_true",
                    help="pyramid workloads: all levels one after the other on the main stream (default: coarse levels forked "
                         "onto side streams and joined inside every step)")
    ap.add_argument("--dry", action="store_true", help="launcher / rendezvous check on the CPU (gloo), no kernels")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 with every rank on cuda:0 and gloo instead of RCCL (which refuses two ranks on one device): the "
                         "whole multi-rank code path with real kernels on a one-GPU box; the line says so and is not a scaling figure")
    ap.add_argument("--no-also", action="store_true",
                    help="default workload at N=1: skip the short joined runs of the other workloads (the `also` object)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))   # before anything here touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.workload == "full_step":     # the config-5 harness itself over gloo, torch stand-ins for the hot path
            from scripts import full_step
            return full_step.main(args, rank, local_rank, world, dry=True)
        return dry_run(args, rank, world)
    if args.workload == "full_step":
        from scripts import full_step
        return full_step.main(args, rank, local_rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import depth_estimation_amd as de
    wl = WORKLOADS[args.workload]
    image_grad = args.image_grad or wl.get("image_grad", False)
    scales = build_problem(de, wl, dev, rank, image_grad, 1)
    # one ordinary (non-null) stream for everything: torch's default stream is HIP's null stream, whose implicit
    # synchronisation with every other stream costs several microseconds per launch at this step size
    main_stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(main_stream)
    st = torch.cuda.current_stream().cuda_stream

    fork_levels = len(scales) > 1 and not args.levels_one_stream
    step_one_stream = make_step(scales, main_stream, dev, False)
    step = make_step(scales, main_stream, dev, fork_levels)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # bring the device to its steady state first: a fresh process starts with the GPU in a low power state and the first
    # ~10 ms of kernels run at lower clocks (measured: 200 steps after 20 warm-up steps read 7 % slower than after 200 or
    # more, profiles/r02_warmup_sensitivity.txt).  Untimed, the same steps, SPIN_UP_S seconds; then the W warm-up steps
    # and the K timed steps of the contract.
    no_spin = None
    if args.spin_up > 0:
        # what the same W + K steps read WITHOUT the spin-up (a fresh process, device in its idle power state)
        for _ in range(args.warmup):
            step()
        barrier()
        n0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        no_spin = time.perf_counter() - n0
        if dist is not None:
            t = torch.tensor([no_spin], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            no_spin = float(t.item())
    s0 = time.perf_counter()
    while time.perf_counter() - s0 < args.spin_up:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    warped_px_rank = sum(sc.units for sc in scales)
    value = world * warped_px_rank * args.steps / elapsed / 1e9
    # N > 1: what the process group itself saw (a SUM of ones over the ranks must read N) and what the only cross-rank
    # traffic a trainer on this path has costs -- one all-reduce of the <= 8 loss / metric scalars (dist.reduce_scalars)
    group = None
    if dist is not None:
        ones = torch.ones(8, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)
        ranks_summed = int(ones[0].item())
        buf = torch.zeros(8, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        group = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks_summed": ranks_summed,
                 "comm_ms": round((time.perf_counter() - c0) / 20 * 1e3, 4),
                 "comm": "all-reduce of 8 fp32 scalars (loss / metric sums); no collective on the data path"}
    levels_one_stream = None
    if fork_levels and rank == 0:   # the same steps with every level on the main stream, for comparison
        for _ in range(args.warmup):
            step_one_stream()
        torch.cuda.synchronize()
        l0 = time.perf_counter()
        for _ in range(args.steps):
            step_one_stream()
        torch.cuda.synchronize()
        ldt = time.perf_counter() - l0
        levels_one_stream = {"value": round(warped_px_rank * args.steps / ldt / 1e9, 4), "unit": "Gpix/s (this rank)",
                             "ms_per_step": round(ldt / args.steps * 1e3, 5)}
    step_bytes = sum(sc.step_bytes for sc in scales)
    kernel_sum_bytes = sum(sum(sc.alg_bytes.values()) for sc in scales)

    # cold-cache figure: the 192x640 working set fits the 256 MiB Infinity Cache, so also time single
    # steps after a 512 MiB write has pushed it out (host clock around one synchronised step: includes ~10 us of
    # launch/sync latency; reported beside `value`, never as it)
    cold = None
    if rank == 0 and args.cold_steps > 0:
        flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        ts = []
        for _ in range(args.cold_steps):
            flush.fill_(1)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - c0)
        del flush
        ts.sort()
        med = ts[len(ts) // 2]
        cold = {"ms_per_step": round(med * 1e3, 5), "value": round(warped_px_rank / med / 1e9, 4), "unit": "Gpix/s",
                "steps": args.cold_steps, "note": "median of single synchronised steps, each after a 512 MiB cache flush"}

    # per-kernel durations: HIP events on the launch stream around each call, inside the same step sequence
    names = scales[0].names
    sc0 = scales[0]
    ev = {n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
          for n in names}
    torch.cuda.synchronize()
    for i in range(args.steps):
        for sc in scales:
            if sc is sc0:
                for n in names:
                    ev[n][i][0].record()
                    sc.calls[n](st)
                    ev[n][i][1].record()
            else:
                sc.step(st)
    torch.cuda.synchronize()
    kern_ms = {n: sum(a.elapsed_time(b) for a, b in ev[n]) / args.steps for n in names}
    dom = max(names, key=lambda n: kern_ms[n])
    achieved = sc0.alg_bytes[dom] / (kern_ms[dom] * 1e-3) / 1e9
    # per launch, from the PMC passes of the same command (profiles/pmc.json: HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE
    # with the gfx950 unit correction; VALU wave instructions = SQ_INSTS_VALU); null if not collected
    pmc = {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as fh:
            pmc = json.load(fh).get(args.workload, {}).get(dom, {})
    except (OSError, ValueError):
        pass
    traffic, valu_insts = pmc.get("hbm_bytes"), pmc.get("valu_wave_insts")
    valu = None
    if valu_insts:
        props = torch.cuda.get_device_properties(dev)
        simds = 4 * props.multi_processor_count
        peak = simds / (VALU_ISSUE_NS * 1e-9) / 1e9          # G wave-instructions / s
        got = valu_insts / (kern_ms[dom] * 1e-3) / 1e9
        valu = {"achieved": round(got, 1), "peak": round(peak, 1), "unit": "G wave-instr/s", "frac": round(got / peak, 4),
                "insts_per_launch": valu_insts, "simds": simds}
        if "valu_fma" in pmc:
            # `peak` is for plain adds.  Issue cost of the counted mix by the per-instruction costs measured in round 1
            # (scripts/isa_cost.py: VOP2 add 2.5 cycles, v_fma 3.0, transcendental 8.2; everything else -- moves, multiplies,
            # compares, selects, VOP3-encoded -- 2.9 on average in these kernels), at the nominal 2.4 GHz
            rest = valu_insts - pmc["valu_fma"] - pmc["valu_add"] - pmc["valu_trans"]
            cyc = 3.0 * pmc["valu_fma"] + 2.5 * pmc["valu_add"] + 8.2 * pmc["valu_trans"] + 2.9 * rest
            valu["mix_issue_ms"] = round(cyc / simds / 2.4e9 * 1e3, 5)
            valu["frac_of_mix_issue_rate"] = round(valu["mix_issue_ms"] / kern_ms[dom], 4)
    # what binds the dominant kernel: the resource it uses the larger fraction of
    hbm_frac = achieved / HBM_PEAK_GBS
    binding = "valu" if (valu is not None and valu["frac"] > hbm_frac) else "hbm"

    # the un-joined variant (round 1's headline): independent sub-batches on their own streams, pipelined across steps
    pipelined = None
    if rank == 0 and args.pipelined > 1 and wl["B"] >= args.pipelined and "sweep" not in wl:
        parts = build_problem(de, wl, dev, rank, image_grad, args.pipelined)
        side = [torch.cuda.Stream() for _ in parts]

        def pstep():
            for j, sc in enumerate(parts):
                sc.step(side[j].cuda_stream)

        n = max(10, args.steps // 2)
        for _ in range(min(args.warmup, 10)):
            pstep()
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        for _ in range(n):
            pstep()
        torch.cuda.synchronize()
        pdt = time.perf_counter() - p0
        pipelined = {"value": round(warped_px_rank * n / pdt / 1e9, 4), "unit": "Gpix/s (this rank)",
                     "ms_per_step": round(pdt / n * 1e3, 5), "steps": n,
                     "launch": f"{args.pipelined} independent sub-batches on {args.pipelined} streams, not joined between steps"}
        del parts

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": METRIC,
            "value": round(value, 4), "unit": "G plane-px/s" if "sweep" in wl else "Gpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "targets_per_gpu": wl["B"], "sources_per_target": wl["S"],
                       "height": wl["H"], "width": wl["W"], "scales": wl["scales"], "image_grad": bool(image_grad),
                       "metrics_reduce": bool(wl.get("metrics", False)),
                       "launch": "eager, steps ordered (every step starts after the previous one has finished); " +
                                 (("one C-ABI call per pyramid level (ds_loss_step_fwdbwd); " +
                                   ("finest level on the main stream, the others forked onto side streams and joined inside every step"
                                    if fork_levels else "all levels on one stream"))
                                  if wl["scales"] > 1 else "whole batch on one stream, " +
                                  ("three C-ABI calls per step (ds_resize_bilinear, ds_plane_sweep_depth_batch, ds_resize_bicubic)"
                                   if "sweep" in wl else "one C-ABI call per step (" +
                                   ("ds_loss_step_fwd: forward only" if wl.get("fwd_only") else "ds_loss_step_fwdbwd") + ")")),
                       "parallelism": f"batch-sharded x{world}, no collective", "spin_up_s": args.spin_up,
                       "pose_sigma": POSE_SIGMA},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(hbm_frac, 4), "traffic": traffic,
                         "alg_bytes_per_launch": sc0.alg_bytes[dom], "avg_ms": round(kern_ms[dom], 5),
                         "valu": valu, "binding_resource": binding},
            "kernels_ms": {n: round(kern_ms[n], 5) for n in names},
            # the step against the HBM roofline by BASELINE.md's accounting (80 B per warped px at S=2, 92 with
            # the image gradient): bytes the fused path must move, not the bytes the unfused kernels do move
            "step_alg_bytes": step_bytes,
            "step_alg_GBps": round(world * step_bytes / (ms * 1e-3) / 1e9, 1),
            "step_frac_of_hbm_peak": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            # for comparison: the sum over the launched kernels of each one's own input+output bytes (includes the
            # warped and g_warped round trips between them)
            "kernel_sum_alg_GBps": round(kernel_sum_bytes / (ms * 1e-3) / 1e9, 1),
        }
        if "sweep" in wl:
            line["config"].update(planes=wl["sweep"]["D"], cost_height=wl["sweep"]["h"], cost_width=wl["sweep"]["w"])
        if no_spin is not None:
            # the same W warm-up + K timed steps BEFORE the spin-up: what a fresh process reads with the device still in
            # its idle power state (the effect of config.spin_up_s on `value`, in the record)
            line["value_no_spin_up"] = round(world * warped_px_rank * args.steps / no_spin / 1e9, 4)
            line["ms_per_step_no_spin_up"] = round(no_spin / args.steps * 1e3, 5)
        if group is not None:
            line["process_group"] = group
        if args.rehearse_on_one_gpu:
            line["rehearsal"] = "all ranks share cuda:0, gloo instead of RCCL: checks the multi-rank path, not a scaling figure"
        if levels_one_stream is not None:
            line["levels_on_one_stream"] = levels_one_stream
        if pipelined is not None:
            line["pipelined_unjoined"] = pipelined
        if cold is not None:
            line["cold_cache"] = cold
        if world == 1 and not args.no_also and args.workload == "b12_s2_192x640_fwdbwd":
            del scales, sc0, step, step_one_stream
            torch.cuda.empty_cache()
            line["also"] = {name: quick_line(de, name, dev, main_stream) for name in ALSO}
        if not args.no_cpu and world == 1:  # the CPU leg is timed at N=1 only
            line["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
