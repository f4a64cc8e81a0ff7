"""The hot path as registered torch operators (`torch.ops.depthsynth.*`): schema, fake (meta) implementation and autograd
formula for each, so that a model calling them traces under torch.compile / torch.export as single graph nodes.

SURVEY.md 8b names a "thin C-ABI torch custom op" as the boundary's first choice.  `ops.py` reaches the same C-ABI calls
through `torch.autograd.Function` (less dispatch overhead per call in eager mode: that stays the default of the
`synthesis` / `metrics` / `fused` facades); this module registers the differentiable ones with `torch.library`:

    depthsynth::pose_from_vec(vec)                                   synthesis.construct_pose_batch  (synthesis.py:6-33)
    depthsynth::warp(K, Kinv, pose, depth, src)                      synthesis.synthesize_from_depth (synthesis.py:40-70)
    depthsynth::reprojection_loss(preds, targets, shared_target)     metrics.reprojection_loss       (metrics.py:22-60)
    depthsynth::loss_step(K, Kinv, pose, depth, src, target, scale)  train.py:72-89 for B targets, value and gradients in
                                                                     one C call (ds_train_step_fwdbwd)

and their backward operators (`*_backward`), which are operators too, so a compiled backward graph holds them as nodes.
Importing this module registers the operators; the functions below call them with the signatures of `ops.py`.
The CUDA-device implementations fail loudly when libdepthsynth.so is missing (as every product path does); there is no
CPU implementation.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor
from torch.library import custom_op

from . import ops as _ops

_DEV = "cuda"


# ---------------------------------------------------------------------------------------------------- pose_from_vec
@custom_op("depthsynth::pose_from_vec", mutates_args=(), device_types=_DEV)
def _pose_from_vec(vec: Tensor) -> Tensor:
    return _ops.pose_from_vec_impl(vec)[1]


@_pose_from_vec.register_fake
def _(vec):
    torch._check(vec.dim() == 2 and vec.shape[1] == 6, lambda: f"pose_vectors must be (N,6), got {tuple(vec.shape)}")
    return vec.new_empty(vec.shape[0], 3, 4, dtype=torch.float32)


@custom_op("depthsynth::pose_from_vec_backward", mutates_args=(), device_types=_DEV)
def _pose_from_vec_backward(vec: Tensor, grad: Tensor) -> Tensor:
    return _ops.pose_from_vec_bwd_impl(_ops.L.dev_f32(vec, "pose_vectors"), grad)


@_pose_from_vec_backward.register_fake
def _(vec, grad):
    return vec.new_empty(vec.shape, dtype=torch.float32)


def _pose_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _pose_backward(ctx, g):
    (vec,) = ctx.saved_tensors
    return _pose_from_vec_backward(vec, g.contiguous())


_pose_from_vec.register_autograd(_pose_backward, setup_context=_pose_setup)


# ---------------------------------------------------------------------------------------------------- warp
@custom_op("depthsynth::warp", mutates_args=(), device_types=_DEV)
def _warp(K: Tensor, Kinv: Tensor, pose: Tensor, depth: Tensor, src: Tensor) -> Tensor:
    return _ops.warp_impl(K, Kinv, pose, depth, src)[1]


@_warp.register_fake
def _(K, Kinv, pose, depth, src):
    torch._check(depth.dim() == 3 and src.dim() == 4 and src.shape[0] == depth.shape[0],
                 lambda: f"depth (N,H,W) and src (N,C,Hs,Ws) expected, got {tuple(depth.shape)}, {tuple(src.shape)}")
    n, h, w = depth.shape
    return src.new_empty(n, src.shape[1], h, w, dtype=torch.float32)


@custom_op("depthsynth::warp_backward", mutates_args=(), device_types=_DEV)
def _warp_backward(K: Tensor, Kinv: Tensor, pose: Tensor, depth: Tensor, src: Tensor, grad: Tensor,
                   need_src: bool) -> Tuple[Tensor, Tensor, Tensor]:
    K, Kinv, pose = _ops._chk_cam(K, Kinv, pose, depth.shape[0])
    depth, src = _ops.L.dev_f32(depth, "depth"), _ops.L.dev_f32(src, "src")
    gp, gd, gs = _ops.warp_bwd_impl(K, Kinv, pose, depth, src, grad, need_src)
    # an operator returns tensors only: without the image gradient the third result is an empty tensor
    return gp, gd, gs if gs is not None else src.new_empty(0)


@_warp_backward.register_fake
def _(K, Kinv, pose, depth, src, grad, need_src):
    return (pose.new_empty(pose.shape, dtype=torch.float32), depth.new_empty(depth.shape, dtype=torch.float32),
            src.new_empty(src.shape if need_src else (0,), dtype=torch.float32))


def _warp_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _warp_bwd(ctx, g):
    K, Kinv, pose, depth, src = ctx.saved_tensors
    need_src = ctx.needs_input_grad[4]
    gp, gd, gs = _warp_backward(K, Kinv, pose, depth, src, g.contiguous(), need_src)
    return None, None, gp, gd, (gs if need_src else None)


_warp.register_autograd(_warp_bwd, setup_context=_warp_setup)


# ---------------------------------------------------------------------------------------------------- reprojection_loss
@custom_op("depthsynth::reprojection_loss", mutates_args=(), device_types=_DEV)
def _reprojection_loss(preds: Tensor, targets: Tensor, shared_target: bool) -> Tuple[Tensor, Tensor]:
    (_, _, aux), _, out = _ops.reprojection_loss_impl(preds, targets, shared_target)
    return out, aux


@_reprojection_loss.register_fake
def _(preds, targets, shared_target):
    torch._check(preds.dim() == 5 and preds.shape[2] == 3, lambda: f"preds must be (B,S,3,H,W), got {tuple(preds.shape)}")
    b, _, _, h, w = preds.shape
    return preds.new_empty(b, h, w, dtype=torch.float32), preds.new_empty(b, h, w, dtype=torch.uint8)


@custom_op("depthsynth::reprojection_loss_backward", mutates_args=(), device_types=_DEV)
def _reprojection_loss_backward(preds: Tensor, targets: Tensor, aux: Tensor, shared_target: bool, grad: Tensor) -> Tensor:
    preds, targets = _ops.L.dev_f32(preds, "preds"), _ops.L.dev_f32(targets, "targets")
    _, s, c, h, w = preds.shape
    strides = (c * h * w, 0) if shared_target else (s * c * h * w, c * h * w)
    return _ops.reprojection_loss_bwd_impl(preds, targets, aux, strides, grad)


@_reprojection_loss_backward.register_fake
def _(preds, targets, aux, shared_target, grad):
    return preds.new_empty(preds.shape, dtype=torch.float32)


def _rl_setup(ctx, inputs, output):
    preds, targets, shared = inputs
    ctx.save_for_backward(preds, targets, output[1])
    ctx.shared = shared
    ctx.mark_non_differentiable(output[1])


def _rl_bwd(ctx, g_out, _g_aux):
    preds, targets, aux = ctx.saved_tensors
    if ctx.needs_input_grad[1]:
        raise RuntimeError("reprojection_loss: gradient w.r.t. targets is not implemented")
    return _reprojection_loss_backward(preds, targets, aux, ctx.shared, g_out.contiguous()), None, None


_reprojection_loss.register_autograd(_rl_bwd, setup_context=_rl_setup)


# ---------------------------------------------------------------------------------------------------- loss_step
@custom_op("depthsynth::loss_step", mutates_args=(), device_types=_DEV)
def _loss_step(K: Tensor, Kinv: Tensor, pose: Tensor, depth: Tensor, src: Tensor, target: Tensor,
               smooth_scale: float) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    return _ops.train_step_fwdbwd_impl(K, Kinv, pose, depth, src, target, smooth_scale)


@_loss_step.register_fake
def _(K, Kinv, pose, depth, src, target, smooth_scale):
    torch._check(src.dim() == 5 and src.shape[2] == 3 and target.dim() == 4,
                 lambda: f"src (B,S,3,H,W) and target (B,3,H,W) expected, got {tuple(src.shape)}, {tuple(target.shape)}")
    b, s, _, h, w = src.shape
    f = dict(dtype=torch.float32)
    return (src.new_empty((), **f), src.new_empty(b, 2, **f), src.new_empty(1, **f), src.new_empty(b, s, 3, h, w, **f),
            src.new_empty(b, h, w, **f), src.new_empty(b, h, w, **f), src.new_empty(b, h, w, dtype=torch.uint8),
            depth.new_empty(depth.shape, **f), src.new_empty(b * s, 3, 4, **f))


def _ls_setup(ctx, inputs, output):
    ctx.pose_shape = inputs[2].shape
    ctx.save_for_backward(output[7], output[8])
    ctx.mark_non_differentiable(*output[1:])
    ctx.set_materialize_grads(False)


def _ls_bwd(ctx, g_total, *_unused):
    if g_total is None:
        return (None,) * 7
    if ctx.needs_input_grad[4] or ctx.needs_input_grad[5]:
        raise RuntimeError("loss_step: gradients w.r.t. the images are not implemented (use warp + reprojection_loss)")
    gd, gp = ctx.saved_tensors
    # total is linear in its upstream gradient: the gradients formed in the forward call are scaled by it
    return None, None, (gp * g_total).reshape(ctx.pose_shape), gd * g_total, None, None, None


_loss_step.register_autograd(_ls_bwd, setup_context=_ls_setup)


# ---------------------------------------------------------------------------------------------------- callers
def pose_from_vec(vec: Tensor) -> Tensor:
    return torch.ops.depthsynth.pose_from_vec(vec)


def warp(K: Tensor, Kinv: Tensor, pose: Tensor, depth: Tensor, src: Tensor) -> Tensor:
    """pose (N,3,4), depth (N,H,W), src (N,C,Hs,Ws) -> warped (N,C,H,W)."""
    return torch.ops.depthsynth.warp(K, Kinv, pose, depth, src)


def reprojection_loss(preds: Tensor, targets: Tensor, shared_target: bool):
    """preds (B,S,3,H,W); targets (B,3,H,W) if shared_target else (B,S,3,H,W) -> (out (B,H,W), aux (B,H,W) u8)."""
    return torch.ops.depthsynth.reprojection_loss(preds, targets, shared_target)


def train_step(K: Tensor, Kinv: Tensor, pose: Tensor, depth: Tensor, src: Tensor, target: Tensor, smooth_scale: float):
    """-> (total, sums (B,2), reg (1), warped, out_loss, idt_loss, aux) as ops.train_step(eager_grads=True)."""
    return torch.ops.depthsynth.loss_step(K, Kinv, pose, depth, src, target, float(smooth_scale))[:7]
