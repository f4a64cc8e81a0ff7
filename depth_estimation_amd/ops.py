"""Batched operators over libdepthsynth.so with torch autograd.

These are the building blocks; `synthesis.py` / `metrics.py` expose them under the reference's names and
call signatures, `fused.py` chains them into the whole reprojection loss of train.py:72-92.
Outputs and workspaces are torch tensors allocated here; the kernels never allocate.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _chk_cam(K, Kinv, pose, n):
    K, Kinv, pose = L.dev_f32(K, "intrinsic_mat"), L.dev_f32(Kinv, "intrinsic_inv"), L.dev_f32(pose, "pose_mat")
    if K.shape != (4, 4) or Kinv.shape != (4, 4):
        raise RuntimeError(f"intrinsics must be (4,4), got {tuple(K.shape)} / {tuple(Kinv.shape)}")
    if pose.shape != (n, 3, 4):
        raise RuntimeError(f"pose must be ({n},3,4), got {tuple(pose.shape)}")
    return K, Kinv, pose


# ----------------------------------------------------------------------------------------------------
# synthesis.construct_pose_batch
# ----------------------------------------------------------------------------------------------------
# (the *_impl functions are the C-ABI calls; the autograd.Function classes below and the torch.library registrations in
# custom_ops.py both wrap them)
def pose_from_vec_impl(vec):
    vec = L.dev_f32(vec, "pose_vectors")
    if vec.dim() != 2 or vec.shape[1] != 6:
        raise RuntimeError(f"pose_vectors must be (N,6), got {tuple(vec.shape)}")
    n = vec.shape[0]
    pose = torch.empty(n, 3, 4, dtype=torch.float32, device=vec.device)
    L.check(L.lib().ds_pose_from_vec_fwd(vec.data_ptr(), n, pose.data_ptr(), L.stream()), "ds_pose_from_vec_fwd")
    return vec, pose


def pose_from_vec_bwd_impl(vec, g):
    g = L.dev_f32(g, "grad")
    gv = torch.empty_like(vec)
    L.check(L.lib().ds_pose_from_vec_bwd(vec.data_ptr(), g.data_ptr(), vec.shape[0], gv.data_ptr(), L.stream()),
            "ds_pose_from_vec_bwd")
    return gv


class _PoseFromVec(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vec):
        vec, pose = pose_from_vec_impl(vec)
        ctx.save_for_backward(vec)
        return pose

    @staticmethod
    def backward(ctx, g):
        (vec,) = ctx.saved_tensors
        return pose_from_vec_bwd_impl(vec, g)


def pose_from_vec(vec: torch.Tensor) -> torch.Tensor:
    return _PoseFromVec.apply(vec)


# ----------------------------------------------------------------------------------------------------
# synthesis.reproject_from_depth (batched)
# ----------------------------------------------------------------------------------------------------
class _Reproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, K, Kinv, pose, depth):
        depth = L.dev_f32(depth, "depth_map")
        n, h, w = depth.shape
        K, Kinv, pose = _chk_cam(K, Kinv, pose, n)
        L.same_device(depth, K, Kinv, pose)
        coords = torch.empty(n, h, w, 2, dtype=torch.float32, device=depth.device)
        L.check(L.lib().ds_reproject(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), n, h, w,
                                     coords.data_ptr(), L.stream()), "ds_reproject")
        ctx.save_for_backward(K, Kinv, pose, depth)
        return coords

    @staticmethod
    def backward(ctx, g):
        K, Kinv, pose, depth = ctx.saved_tensors
        n, h, w = depth.shape
        g = L.dev_f32(g, "grad")
        gd = torch.empty_like(depth)
        gp = torch.empty_like(pose)
        nb = L.lib().ds_warp_bwd_workspace_bytes(n, h, w)
        ws = L.workspace.get(nb, depth.device)
        L.check(L.lib().ds_reproject_bwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), g.data_ptr(), n,
                                         h, w, gd.data_ptr(), gp.data_ptr(), ws.data_ptr(), nb, L.stream()),
                "ds_reproject_bwd")
        return None, None, gp, gd


def reproject(K, Kinv, pose, depth) -> torch.Tensor:
    """pose (N,3,4), depth (N,H,W) -> (N,H,W,2) pixel coords."""
    return _Reproject.apply(K, Kinv, pose, depth)


# ----------------------------------------------------------------------------------------------------
# synthesis.synthesize (grid_sample with in-place normalisation of the grid)
# ----------------------------------------------------------------------------------------------------
class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, grid):
        # src (N,C,Hs,Ws) contiguous, grid (N,H,W,2) contiguous, mutated in place
        n, c, hs, ws = src.shape
        _, h, w, _ = grid.shape
        out = torch.empty(n, c, h, w, dtype=torch.float32, device=src.device)
        L.check(L.lib().ds_grid_sample_fwd(src.data_ptr(), grid.data_ptr(), n, c, hs, ws, h, w, out.data_ptr(), L.stream()),
                "ds_grid_sample_fwd")
        ctx.mark_dirty(grid)
        ctx.save_for_backward(src, grid)
        ctx.src_needs_grad = src.requires_grad
        return out, grid

    @staticmethod
    def backward(ctx, g_out, _g_grid_unused):
        src, grid = ctx.saved_tensors
        n, c, hs, ws = src.shape
        _, h, w, _ = grid.shape
        g_out = L.dev_f32(g_out, "grad")
        g_grid = torch.empty_like(grid)
        g_src = torch.zeros_like(src) if ctx.needs_input_grad[0] else None
        L.check(L.lib().ds_grid_sample_bwd(src.data_ptr(), grid.data_ptr(), g_out.data_ptr(), n, c, hs, ws, h, w,
                                           g_grid.data_ptr(), L.ptr(g_src), L.stream()), "ds_grid_sample_bwd")
        if _g_grid_unused is not None:
            # gradient arriving through later uses of the (now normalised) grid: chain rule of the in-place
            # normalisation gx = u*2/(Ws-1) - 1
            scale = torch.tensor([2.0 / (ws - 1), 2.0 / (hs - 1)], dtype=torch.float32, device=grid.device)
            g_grid = g_grid + _g_grid_unused * scale
        return g_src, g_grid


def grid_sample_(src: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """In-place-normalising sampler: `grid` (N,H,W,2), contiguous, holds pixel coords and is overwritten."""
    out, _ = _GridSample.apply(src, grid)
    return out


# ----------------------------------------------------------------------------------------------------
# fused reproject + sample
# ----------------------------------------------------------------------------------------------------
def warp_impl(K, Kinv, pose, depth, src):
    depth, src = L.dev_f32(depth, "depth"), L.dev_f32(src, "src")
    n, h, w = depth.shape
    if src.dim() != 4 or src.shape[0] != n:
        raise RuntimeError(f"src must be ({n},C,Hs,Ws), got {tuple(src.shape)}")
    _, c, hs, ws = src.shape
    K, Kinv, pose = _chk_cam(K, Kinv, pose, n)
    L.same_device(depth, src, K, Kinv, pose)
    warped = torch.empty(n, c, h, w, dtype=torch.float32, device=src.device)
    L.check(L.lib().ds_warp_fwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(), n, c,
                                hs, ws, h, w, warped.data_ptr(), L.stream()), "ds_warp_fwd")
    return (K, Kinv, pose, depth, src), warped


def warp_bwd_impl(K, Kinv, pose, depth, src, g, need_src: bool):
    """-> (g_pose, g_depth, g_src or None); the arguments are warp_impl's checked tensors."""
    n, h, w = depth.shape
    _, c, hs, ws = src.shape
    g = L.dev_f32(g, "grad")
    gd = torch.empty_like(depth)
    gp = torch.empty_like(pose)
    gs = None
    if need_src:
        # C = 3: every texel is written (no float atomics, run-to-run identical); other C: atomics onto zeros
        gs = torch.empty_like(src) if c == 3 else torch.zeros_like(src)
        nb = L.lib().ds_warp_bwd_image_workspace_bytes(n, c, hs, ws, h, w)
    else:
        nb = L.lib().ds_warp_bwd_workspace_bytes(n, h, w)
    wsb = L.workspace.get(nb, depth.device)
    L.check(L.lib().ds_warp_bwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                g.data_ptr(), n, c, hs, ws, h, w, gd.data_ptr(), gp.data_ptr(), L.ptr(gs),
                                wsb.data_ptr(), nb, L.stream()), "ds_warp_bwd")
    return gp, gd, gs


class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, K, Kinv, pose, depth, src):
        saved, warped = warp_impl(K, Kinv, pose, depth, src)
        ctx.save_for_backward(*saved)
        return warped

    @staticmethod
    def backward(ctx, g):
        gp, gd, gs = warp_bwd_impl(*ctx.saved_tensors, g, ctx.needs_input_grad[4])
        return None, None, gp, gd, gs


def warp(K, Kinv, pose, depth, src) -> torch.Tensor:
    """pose (N,3,4), depth (N,H,W), src (N,C,Hs,Ws) -> warped (N,C,H,W)."""
    return _Warp.apply(K, Kinv, pose, depth, src)


# ----------------------------------------------------------------------------------------------------
# metrics.reprojection_error
# ----------------------------------------------------------------------------------------------------
class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, targets):
        preds, targets = L.dev_f32(preds, "preds"), L.dev_f32(targets, "targets")
        if preds.shape != targets.shape:
            raise RuntimeError(f"preds {tuple(preds.shape)} and targets {tuple(targets.shape)} must have the same shape")
        L.same_device(preds, targets)
        h, w = preds.shape[-2:]
        out = torch.empty_like(preds)
        L.check(L.lib().ds_ssim_fwd(preds.data_ptr(), targets.data_ptr(), preds.numel() // (h * w), h, w, out.data_ptr(),
                                    L.stream()), "ds_ssim_fwd")
        ctx.save_for_backward(preds, targets)
        return out

    @staticmethod
    def backward(ctx, g):
        preds, targets = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise RuntimeError("reprojection_error: gradient w.r.t. targets is not implemented "
                               "(no caller in the reference needs it)")
        h, w = preds.shape[-2:]
        g = L.dev_f32(g, "grad")
        gp = torch.empty_like(preds)
        L.check(L.lib().ds_ssim_bwd(preds.data_ptr(), targets.data_ptr(), g.data_ptr(), preds.numel() // (h * w), h, w,
                                    gp.data_ptr(), L.stream()), "ds_ssim_bwd")
        return gp, None


def ssim_loss(preds, targets) -> torch.Tensor:
    return _Ssim.apply(preds, targets)


# ----------------------------------------------------------------------------------------------------
# metrics.reprojection_loss (batched: preds (B,S,3,H,W))
# ----------------------------------------------------------------------------------------------------
def reprojection_loss_impl(preds, targets, shared_target: bool):
    preds, targets = L.dev_f32(preds, "preds"), L.dev_f32(targets, "targets")
    L.same_device(preds, targets)
    b, s, c, h, w = preds.shape
    if c != 3:
        raise RuntimeError(f"reprojection_loss kernels are built for 3 channels, got {c}")
    hw = h * w
    if shared_target:
        if targets.shape != (b, c, h, w):
            raise RuntimeError(f"targets must be ({b},{c},{h},{w}), got {tuple(targets.shape)}")
        sb, ss = c * hw, 0
    else:
        if targets.shape != preds.shape:
            raise RuntimeError(f"targets {tuple(targets.shape)} must match preds {tuple(preds.shape)}")
        sb, ss = s * c * hw, c * hw
    out = torch.empty(b, h, w, dtype=torch.float32, device=preds.device)
    aux = torch.empty(b, h, w, dtype=torch.uint8, device=preds.device)
    L.check(L.lib().ds_reprojection_loss_fwd(preds.data_ptr(), targets.data_ptr(), sb, ss, b, s, h, w, out.data_ptr(),
                                             aux.data_ptr(), L.stream()), "ds_reprojection_loss_fwd")
    return (preds, targets, aux), (sb, ss), out


def reprojection_loss_bwd_impl(preds, targets, aux, strides, g_out):
    b, s, c, h, w = preds.shape
    sb, ss = strides
    g_out = L.dev_f32(g_out, "grad")
    gp = torch.empty_like(preds)
    L.check(L.lib().ds_reprojection_loss_bwd(preds.data_ptr(), targets.data_ptr(), sb, ss, g_out.data_ptr(),
                                             aux.data_ptr(), b, s, h, w, gp.data_ptr(), L.stream()),
            "ds_reprojection_loss_bwd")
    return gp


class _ReprojLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, preds, targets, shared_target: bool):
        saved, ctx.strides, out = reprojection_loss_impl(preds, targets, shared_target)
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(saved[2])
        return out, saved[2]

    @staticmethod
    def backward(ctx, g_out, _g_aux):
        preds, targets, aux = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise RuntimeError("reprojection_loss: gradient w.r.t. targets is not implemented")
        return reprojection_loss_bwd_impl(preds, targets, aux, ctx.strides, g_out), None, None


def reprojection_loss(preds, targets, shared_target: bool):
    """preds (B,S,3,H,W); targets (B,3,H,W) if shared_target else (B,S,3,H,W) -> (out (B,H,W), aux (B,H,W) u8)."""
    return _ReprojLoss.apply(preds, targets, shared_target)


# ----------------------------------------------------------------------------------------------------
# loss assembly train.py:80-92
# ----------------------------------------------------------------------------------------------------
class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, warped, src, target):
        warped, src, target = L.dev_f32(warped, "warped"), L.dev_f32(src, "src"), L.dev_f32(target, "target")
        L.same_device(warped, src, target)
        b, s, c, h, w = warped.shape
        if c != 3 or src.shape != warped.shape or target.shape != (b, c, h, w):
            raise RuntimeError(f"expected warped/src (B,S,3,H,W) and target (B,3,H,W); got {tuple(warped.shape)}, "
                               f"{tuple(src.shape)}, {tuple(target.shape)}")
        dev = warped.device
        out = torch.empty(b, h, w, dtype=torch.float32, device=dev)
        idt = torch.empty(b, h, w, dtype=torch.float32, device=dev)
        aux = torch.empty(b, h, w, dtype=torch.uint8, device=dev)
        sums = torch.empty(b, 2, dtype=torch.float32, device=dev)
        nb = L.lib().ds_fused_loss_workspace_bytes(b, h, w)
        ws = L.workspace.get(nb, dev)
        L.check(L.lib().ds_fused_loss_fwd(warped.data_ptr(), src.data_ptr(), target.data_ptr(), b, s, h, w, out.data_ptr(),
                                          idt.data_ptr(), aux.data_ptr(), sums.data_ptr(), ws.data_ptr(), nb, L.stream()),
                "ds_fused_loss_fwd")
        ctx.save_for_backward(warped, target, idt, aux)
        ctx.mark_non_differentiable(out, idt, aux)
        ctx.set_materialize_grads(False)  # no zero-filled (B,H,W) gradients for the three maps
        return sums, out, idt, aux

    @staticmethod
    def backward(ctx, g_sums, _g_out, _g_idt, _g_aux):
        if g_sums is None:
            return None, None, None
        warped, target, idt, aux = ctx.saved_tensors
        b, s, c, h, w = warped.shape
        # only sums[:,0] (the auto-masked mean, train.py:88) carries gradient; sums[:,1] is a log value
        g0 = L.dev_f32(g_sums[:, 0], "grad")
        gw = torch.empty_like(warped)
        L.check(L.lib().ds_fused_loss_bwd(warped.data_ptr(), target.data_ptr(), idt.data_ptr(), aux.data_ptr(),
                                          g0.data_ptr(), b, s, h, w, gw.data_ptr(), L.stream()), "ds_fused_loss_bwd")
        return gw, None, None


def fused_loss(warped, src, target):
    """-> (sums (B,2), out_loss (B,H,W), idt_loss (B,H,W), aux (B,H,W) u8).  Differentiable through sums[:,0]
    w.r.t. warped (the logged mean sums[:,1] is detached in the reference too: .item(), train.py:92)."""
    return _FusedLoss.apply(warped, src, target)


# ----------------------------------------------------------------------------------------------------
# metrics.regularization_term
# ----------------------------------------------------------------------------------------------------
class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depths, targets):
        depths, targets = L.dev_f32(depths, "depths"), L.dev_f32(targets, "target_imgs")
        L.same_device(depths, targets)
        b, c, h, w = targets.shape
        n = depths.shape[0]
        if depths.dim() != 4 or depths.shape[1:] != (1, h, w) or n == 0 or n % b:
            raise RuntimeError(f"depths must be (k*{b},1,{h},{w}), got {tuple(depths.shape)}")
        s = n // b
        val = torch.empty(1, dtype=torch.float32, device=depths.device)
        nb = L.lib().ds_smooth_workspace_bytes(n, h, w)
        ws = L.workspace.get(nb, depths.device)
        L.check(L.lib().ds_smooth_batch_fwd(depths.data_ptr(), targets.data_ptr(), b, s, c, h, w, val.data_ptr(),
                                            ws.data_ptr(), nb, L.stream()), "ds_smooth_batch_fwd")
        ctx.save_for_backward(depths, targets)
        return val.reshape(())

    @staticmethod
    def backward(ctx, g):
        depths, targets = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise RuntimeError("regularization_term: gradient w.r.t. target_imgs is not implemented")
        b, c, h, w = targets.shape
        s = depths.shape[0] // b
        g = L.dev_f32(g.reshape(1), "grad")
        gd = torch.empty_like(depths)
        L.check(L.lib().ds_smooth_batch_bwd(depths.data_ptr(), targets.data_ptr(), g.data_ptr(), b, s, c, h, w, 0,
                                            gd.data_ptr(), L.stream()), "ds_smooth_batch_bwd")
        return gd, None


def smoothness(depths, targets) -> torch.Tensor:
    """metrics.regularization_term (metrics.py:75-97): depths (N,1,H,W), targets (N,C,H,W) -> scalar.  With targets
    (B,C,H,W) and N = B*S the S maps b*S..b*S+S-1 share target b (the stack of train.py:80 left implicit)."""
    return _Smooth.apply(depths, targets)


# ----------------------------------------------------------------------------------------------------
# the whole loss step as one autograd node (train.py:72-89 for B targets)
# ----------------------------------------------------------------------------------------------------
class _TrainStep(torch.autograd.Function):
    """Two C-ABI calls forward (warp + photometric loss, smoothness) and two backward, no intermediate
    autograd nodes, no elementwise torch kernels over image-sized tensors: the smoothness gradient is accumulated onto
    the photometric depth gradient inside its kernel."""

    @staticmethod
    def forward(ctx, K, Kinv, pose, depth, src, target, smooth_scale):
        depth, src, target = L.dev_f32(depth, "depth"), L.dev_f32(src, "src"), L.dev_f32(target, "target")
        b, s, c, hs, ws = src.shape
        h, w = target.shape[-2:]
        n = b * s
        ctx.pose_shape = pose.shape
        if c != 3 or target.shape != (b, 3, h, w) or depth.numel() != n * h * w or (hs, ws) != (h, w):
            raise RuntimeError(f"expected src (B,S,3,H,W), target (B,3,H,W), depth (B,S,1,H,W); got {tuple(src.shape)}, "
                               f"{tuple(target.shape)}, {tuple(depth.shape)}")
        K, Kinv, pose = _chk_cam(K, Kinv, pose.reshape(n, 3, 4), n)
        L.same_device(depth, src, target, K, Kinv, pose)
        dev, lib, st = src.device, L.lib(), L.stream()
        f32 = dict(dtype=torch.float32, device=dev)
        nb = max(lib.ds_loss_step_workspace_bytes(b, s, h, w), lib.ds_smooth_workspace_bytes(n, h, w))
        wsb = L.workspace.get(nb, dev)
        warped = torch.empty(b, s, 3, h, w, **f32)
        out, idt = torch.empty(b, h, w, **f32), torch.empty(b, h, w, **f32)
        aux = torch.empty(b, h, w, dtype=torch.uint8, device=dev)
        sums, reg = torch.empty(b, 2, **f32), torch.empty(1, **f32)
        L.check(lib.ds_loss_step_fwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                     target.data_ptr(), b, s, h, w, warped.data_ptr(), out.data_ptr(), idt.data_ptr(),
                                     aux.data_ptr(), sums.data_ptr(), wsb.data_ptr(), nb, st),
                "ds_loss_step_fwd")
        L.check(lib.ds_smooth_batch_fwd(depth.data_ptr(), target.data_ptr(), b, s, 3, h, w, reg.data_ptr(), wsb.data_ptr(), nb,
                                        st), "ds_smooth_batch_fwd")
        total = sums[:, 0].sum() + smooth_scale * reg[0]
        ctx.save_for_backward(K, Kinv, pose, depth, src, target, warped, idt, aux)
        ctx.smooth_scale, ctx.nb = smooth_scale, nb
        ctx.mark_non_differentiable(sums, reg, warped, out, idt, aux)
        ctx.set_materialize_grads(False)
        return total, sums, reg, warped, out, idt, aux

    @staticmethod
    def backward(ctx, g_total, *_unused):
        if g_total is None:
            return (None,) * 7
        K, Kinv, pose, depth, src, target, warped, idt, aux = ctx.saved_tensors
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            raise RuntimeError("train_step_loss: gradients w.r.t. the images are not implemented (use "
                               "fused_reprojection_loss(image_grad=True))")
        b, s, _, h, w = src.shape
        dev, lib, st = src.device, L.lib(), L.stream()
        g_total = L.dev_f32(g_total.reshape(1), "grad")
        g_sums = g_total.expand(b).contiguous()
        g_reg = g_total * ctx.smooth_scale
        gw = torch.empty_like(warped)
        gd, gp = torch.empty_like(depth), torch.empty_like(pose)
        wsb = L.workspace.get(ctx.nb, dev)
        L.check(lib.ds_loss_step_bwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                     target.data_ptr(), warped.data_ptr(), idt.data_ptr(), aux.data_ptr(), g_sums.data_ptr(),
                                     b, s, h, w, gw.data_ptr(), gd.data_ptr(), gp.data_ptr(), wsb.data_ptr(), ctx.nb, st),
                "ds_loss_step_bwd")
        L.check(lib.ds_smooth_batch_bwd(depth.data_ptr(), target.data_ptr(), g_reg.data_ptr(), b, s, 3, h, w, 1,
                                        gd.data_ptr(), st), "ds_smooth_batch_bwd")
        return None, None, gp.reshape(ctx.pose_shape), gd, None, None, None


def train_step_fwdbwd_impl(K, Kinv, pose, depth, src, target, smooth_scale: float):
    """ONE C call (ds_train_step_fwdbwd) -> (total, sums, reg, warped, out, idt, aux, g_depth, g_pose (B*S,3,4))."""
    depth, src, target = L.dev_f32(depth, "depth"), L.dev_f32(src, "src"), L.dev_f32(target, "target")
    b, s, c, hs, ws = src.shape
    h, w = target.shape[-2:]
    n = b * s
    if c != 3 or target.shape != (b, 3, h, w) or depth.numel() != n * h * w or (hs, ws) != (h, w):
        raise RuntimeError(f"expected src (B,S,3,H,W), target (B,3,H,W), depth (B,S,1,H,W); got {tuple(src.shape)}, "
                           f"{tuple(target.shape)}, {tuple(depth.shape)}")
    K, Kinv, pose = _chk_cam(K, Kinv, pose.reshape(n, 3, 4), n)
    L.same_device(depth, src, target, K, Kinv, pose)
    dev, lib, st = src.device, L.lib(), L.stream()
    f32 = dict(dtype=torch.float32, device=dev)
    nb = lib.ds_train_step_fwdbwd_workspace_bytes(b, s, h, w)
    wsb = L.workspace.get(nb, dev)
    warped = torch.empty(b, s, 3, h, w, **f32)
    # d loss / d warped stays inside the kernel for S <= 3; only the S = 4 path needs the (B,S,3,H,W) scratch
    gw = torch.empty(b, s, 3, h, w, **f32) if s > 3 else None
    out, idt = torch.empty(b, h, w, **f32), torch.empty(b, h, w, **f32)
    aux = torch.empty(b, h, w, dtype=torch.uint8, device=dev)
    sums, reg, total = torch.empty(b, 2, **f32), torch.empty(1, **f32), torch.empty((), **f32)
    gd, gp = torch.empty_like(depth), torch.empty(n, 3, 4, **f32)
    L.check(lib.ds_train_step_fwdbwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                     target.data_ptr(), smooth_scale, b, s, h, w, warped.data_ptr(), out.data_ptr(),
                                     idt.data_ptr(), aux.data_ptr(), sums.data_ptr(), reg.data_ptr(), total.data_ptr(),
                                     L.ptr(gw), gd.data_ptr(), gp.data_ptr(), wsb.data_ptr(), nb, st),
            "ds_train_step_fwdbwd")
    return total, sums, reg, warped, out, idt, aux, gd, gp


class _TrainStepEager(torch.autograd.Function):
    """The same step with its gradients formed in the forward call: ONE C call, ds_train_step_fwdbwd (the photometric step's
    three launches, the smoothness value and gradient from one kernel, a finalize that also forms total).  total is linear
    in its upstream gradient, so backward only scales the stored gradients by it (loss.backward() passes 1)."""

    @staticmethod
    def forward(ctx, K, Kinv, pose, depth, src, target, smooth_scale):
        ctx.pose_shape = pose.shape
        total, sums, reg, warped, out, idt, aux, gd, gp = train_step_fwdbwd_impl(K, Kinv, pose, depth, src, target, smooth_scale)
        ctx.save_for_backward(gd, gp)
        ctx.mark_non_differentiable(sums, reg, warped, out, idt, aux)
        ctx.set_materialize_grads(False)
        return total, sums, reg, warped, out, idt, aux

    @staticmethod
    def backward(ctx, g_total, *_unused):
        if g_total is None:
            return (None,) * 7
        if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
            raise RuntimeError("train_step_loss: gradients w.r.t. the images are not implemented (use "
                               "fused_reprojection_loss(image_grad=True))")
        gd, gp = ctx.saved_tensors
        return None, None, (gp * g_total).reshape(ctx.pose_shape), gd * g_total, None, None, None


def train_step(K, Kinv, pose, depth, src, target, smooth_scale: float, eager_grads: bool = False):
    """-> (total, sums (B,2), reg (1), warped, out_loss, idt_loss, aux): total = sum_b sums[b,0] + smooth_scale *
    regularization_term(depth maps, targets), differentiable w.r.t. depth and pose; pose (B*S,3,4) or (B,S,3,4)."""
    fn = _TrainStepEager if eager_grads else _TrainStep
    return fn.apply(K, Kinv, pose, depth, src, target, float(smooth_scale))


# ----------------------------------------------------------------------------------------------------
# forward-only ops
# ----------------------------------------------------------------------------------------------------
def depth_metrics(pred, gt, want_counts: bool = False, eps: float = 1e-6):
    """pred, gt (B,1,H,W) -> (8,) fp32: rmse, rmsle, abs_rel, sq_rel, delta1, delta2, delta3, n_valid; with want_counts
    also (B,4) int64: per image the exact counts under the three thresholds and of valid pixels.  eps: the offset inside
    rmsle's logarithms (metrics.py:114)."""
    pred, gt = L.dev_f32(pred.detach(), "pred_depths"), L.dev_f32(gt.detach(), "depths")
    L.same_device(pred, gt)
    if pred.shape != gt.shape or pred.dim() < 3:
        raise RuntimeError(f"pred {tuple(pred.shape)} and gt {tuple(gt.shape)} must both be (B,1,H,W)")
    b = pred.shape[0]
    hw = pred.numel() // b
    h, w = pred.shape[-2], hw // pred.shape[-2]
    out = torch.empty(8, dtype=torch.float32, device=pred.device)
    counts = torch.empty(b, 4, dtype=torch.int64, device=pred.device) if want_counts else None
    nb = L.lib().ds_metrics_workspace_bytes(b, h, w)
    ws = L.workspace.get(nb, pred.device)
    L.check(L.lib().ds_depth_metrics_eps(pred.data_ptr(), gt.data_ptr(), b, h, w, float(eps), out.data_ptr(), L.ptr(counts),
                                         ws.data_ptr(), nb, L.stream()), "ds_depth_metrics_eps")
    return (out, counts) if want_counts else out


def _depths_arg(depths):
    d = np.ascontiguousarray(np.asarray(depths, dtype=np.float64).reshape(-1))
    return d, d.ctypes.data_as(C.POINTER(C.c_double))


def synthesize_from_depths(K, Kinv, pose, depths, width, height, src) -> torch.Tensor:
    src = L.dev_f32(src.detach(), "src_img")
    if src.dim() == 4 and src.shape[0] == 1:
        src = src[0]
    if src.dim() != 3:
        raise RuntimeError(f"src_img must be (C,H,W), got {tuple(src.shape)}")
    c, hs, ws = src.shape
    K, Kinv, pose = _chk_cam(K.detach(), Kinv.detach(), pose.detach().reshape(1, 3, 4), 1)
    L.same_device(src, K, Kinv, pose)
    d, dp = _depths_arg(depths)
    out = torch.empty(len(d), c, int(height), int(width), dtype=torch.float32, device=src.device)
    L.check(L.lib().ds_synthesize_from_depths(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), dp, len(d), src.data_ptr(), c,
                                              hs, ws, int(height), int(width), out.data_ptr(), L.stream()),
            "ds_synthesize_from_depths")
    return out


def cost_volume(target, synth) -> torch.Tensor:
    target, synth = L.dev_f32(target.detach(), "target_img"), L.dev_f32(synth.detach(), "synthesized_imgs")
    L.same_device(target, synth)
    d, c, h, w = synth.shape
    if target.numel() != c * h * w:
        raise RuntimeError(f"target_img must be ({c},{h},{w}), got {tuple(target.shape)}")
    out = torch.empty(d, h, w, dtype=torch.float32, device=synth.device)
    L.check(L.lib().ds_cost_volume(target.data_ptr(), synth.data_ptr(), d, c, h, w, out.data_ptr(), L.stream()),
            "ds_cost_volume")
    return out


def depth_from_costs(cost, sid: bool) -> torch.Tensor:
    cost = L.dev_f32(cost.detach(), "cost_volumes")
    n, d, h, w = cost.shape
    out = torch.empty(n, 1, h, w, dtype=torch.float32, device=cost.device)
    L.check(L.lib().ds_depth_from_costs(cost.data_ptr(), n, d, h, w, int(bool(sid)), out.data_ptr(), L.stream()),
            "ds_depth_from_costs")
    return out


def plane_sweep_depth(K, Kinv, pose, depths, src, target_small, sid: bool = True) -> torch.Tensor:
    """pose (NS,3,4), src (NS,C,Hs,Ws), target_small (C,H,W) -> candidate depth (NS,1,H,W) (train.py:49-59)."""
    src, target_small = L.dev_f32(src.detach(), "src"), L.dev_f32(target_small.detach(), "target_small")
    ns, c, hs, ws = src.shape
    if target_small.dim() == 4:
        target_small = target_small[0]
    _, h, w = target_small.shape
    K, Kinv, pose = _chk_cam(K.detach(), Kinv.detach(), pose.detach(), ns)
    L.same_device(src, target_small, K, Kinv, pose)
    d, dp = _depths_arg(depths)
    out = torch.empty(ns, 1, h, w, dtype=torch.float32, device=src.device)
    L.check(L.lib().ds_plane_sweep_depth(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), dp, len(d), src.data_ptr(),
                                         target_small.data_ptr(), ns, c, hs, ws, h, w, int(bool(sid)), out.data_ptr(),
                                         L.stream()), "ds_plane_sweep_depth")
    return out


def plane_sweep_depth_batch(K, Kinv, pose, depths, src, targets_small, sid: bool = True) -> torch.Tensor:
    """The loop over the batch of train.py:46-63 in one launch: pose (B,S,3,4), src (B,S,C,Hs,Ws), targets_small
    (B,C,H,W) -> candidate depth (B,S,1,H,W); row (b, s) equals plane_sweep_depth(..., src[b], targets_small[b])[s]."""
    src, targets_small = L.dev_f32(src.detach(), "src"), L.dev_f32(targets_small.detach(), "targets_small")
    if src.dim() != 5 or targets_small.dim() != 4 or src.shape[0] != targets_small.shape[0]:
        raise ValueError("src must be (B,S,C,Hs,Ws) and targets_small (B,C,H,W) with the same B")
    b, s_, c, hs, ws = src.shape
    if targets_small.shape[1] != c:
        raise ValueError("src and targets_small differ in channels")
    _, _, h, w = targets_small.shape
    K, Kinv, pose = _chk_cam(K.detach(), Kinv.detach(), pose.detach().reshape(b * s_, 3, 4), b * s_)
    L.same_device(src, targets_small, K, Kinv, pose)
    d, dp = _depths_arg(depths)
    out = torch.empty(b, s_, 1, h, w, dtype=torch.float32, device=src.device)
    L.check(L.lib().ds_plane_sweep_depth_batch(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), dp, len(d),
                                               src.data_ptr(), targets_small.data_ptr(), b, s_, c, hs, ws, h, w,
                                               int(bool(sid)), out.data_ptr(), L.stream()), "ds_plane_sweep_depth_batch")
    return out


# ----------------------------------------------------------------------------------------------------
# functional.interpolate as train.py:55 / :60 call it
# ----------------------------------------------------------------------------------------------------
def _resize(name: str, x: torch.Tensor, size) -> torch.Tensor:
    x = L.dev_f32(x.detach(), "x")
    if x.dim() < 2:
        raise ValueError("x must be (...,H,W)")
    ho, wo = int(size[0]), int(size[1])
    hi, wi = x.shape[-2:]
    n = x.numel() // (hi * wi) if hi * wi else 0
    out = torch.empty(*x.shape[:-2], ho, wo, dtype=torch.float32, device=x.device)
    L.check(getattr(L.lib(), name)(x.data_ptr(), n, hi, wi, ho, wo, out.data_ptr(), L.stream()), name)
    return out


def resize_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    """functional.interpolate(x, size=size, mode="bilinear") (train.py:55); x (...,H,W), no gradient."""
    return _resize("ds_resize_bilinear", x, size)


def resize_bicubic(x: torch.Tensor, size) -> torch.Tensor:
    """functional.interpolate(x, size=size, mode="bicubic") (train.py:60); x (...,H,W), no gradient."""
    return _resize("ds_resize_bicubic", x, size)
