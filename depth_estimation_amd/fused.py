"""Batched, fused entry points: the whole of train.py:72-92 for B targets x S sources in a handful of
kernel launches, with autograd to depth, pose (and optionally the source images)."""
from __future__ import annotations

import torch

from . import ops


def intrinsics(width: int, height: int, device=None):
    """The reference's camera (constants.py:25-44): normalised K with rows scaled by (W,H), and its pinv."""
    K = torch.tensor([[0.58, 0.00, 0.50, 0.00], [0.00, 1.92, 0.50, 0.00], [0.00, 0.00, 1.00, 0.00],
                      [0.00, 0.00, 0.00, 1.00]], dtype=torch.float32)
    K[0, :] *= width
    K[1, :] *= height
    Kinv = torch.linalg.pinv(K)
    return K.to(device), Kinv.to(device)


def fused_reprojection_loss(src: torch.Tensor, target: torch.Tensor, depth: torch.Tensor, pose: torch.Tensor,
                            K: torch.Tensor, Kinv: torch.Tensor, image_grad: bool = False):
    """src (B,S,3,Hs,Ws), target (B,3,H,W), depth (B,S,1,H,W) [metres], pose (B,S,3,4).

    Returns dict(loss (B,) = mean(out/exp(idt+1e-6)) per target (train.py:88), log (B,) = mean(out)
    (train.py:92), warped (B,S,3,H,W), out_loss, idt_loss (B,H,W), aux (B,H,W) u8).
    `loss` is differentiable w.r.t. depth and pose (and src through the warp when image_grad=True; the
    identity term never back-propagates, as in the reference where src carries no grad)."""
    b, s = src.shape[:2]
    c, hs, ws = src.shape[2:]
    h, w = target.shape[-2:]
    src_w = src if image_grad else src.detach()
    warped = ops.warp(K, Kinv, pose.reshape(b * s, 3, 4), depth.reshape(b * s, h, w), src_w.reshape(b * s, c, hs, ws))
    warped = warped.reshape(b, s, c, h, w)
    sums, out_loss, idt_loss, aux = ops.fused_loss(warped, src.detach(), target)
    return dict(loss=sums[:, 0], log=sums[:, 1], warped=warped, out_loss=out_loss, idt_loss=idt_loss, aux=aux)


def train_step_loss(src, target, depth, pose, K, Kinv, max_depth: float = 80.0, smooth_weight: float = 0.02,
                    eager_grads=None):
    """Summed training loss of train.py:83-89 over the B targets: sum_b [ mean(out/exp(idt+1e-6)) +
    0.02 * regularization_term(depth_b/80, target_b stacked S times) ], as ONE autograd node (ops.train_step).

    The reference's regularization mean is per sample (N = S maps of H*W pixels each, train.py:83) and the samples are
    summed (train.py:89): every sample has the same pixel count, so the sum of the per-sample means is B x the mean over
    all B*S maps -- one launch instead of B, and the S maps of a target share its edge weights.  The term is linear in
    the depth scale, regularization_term(d / 80) = regularization_term(d) / 80, which saves an elementwise pass over the
    depth maps and its backward.

    eager_grads=True forms the depth and pose gradients inside the forward call (ds_train_step_fwdbwd: ONE C call for the
    photometric step, the smoothness term and the total); backward then only scales them by its upstream gradient.  For a
    training loop that always calls backward (train.py:103) it is the faster form (0.169 against 0.225 ms at B=12
    192x640); a forward-only caller should not pay for gradients.  None (default): eager exactly when autograd is
    recording and depth or pose requires a gradient -- the only differentiable output is `total` either way."""
    if eager_grads is None:
        eager_grads = torch.is_grad_enabled() and (depth.requires_grad or pose.requires_grad)
    b = src.shape[0]
    total, sums, reg, warped, out_loss, idt_loss, aux = ops.train_step(K, Kinv, pose, depth, src.detach(), target.detach(),
                                                                       smooth_weight * b / max_depth, eager_grads)
    return dict(total=total, loss=sums[:, 0], log=sums[:, 1], reg=reg, warped=warped, out_loss=out_loss,
                idt_loss=idt_loss, aux=aux)


def candidate_depths(src, target, pose, K_cost, Kinv_cost, depths, cost_size, sid: bool = True) -> torch.Tensor:
    """train.py:49-60 for a whole batch in three launches: the target at cost-volume resolution (bilinear, :55), the
    plane sweep with its soft arg-min (:49-59), the candidate depth back at image resolution (bicubic, :60).
    src (B,S,3,H,W), target (B,3,H,W), pose (B,S,3,4), K_cost/Kinv_cost the intrinsics of the cost grid
    (constants.py:33-37), cost_size (COST_HEIGHT, COST_WIDTH) -> (B,S,1,H,W).  No gradient flows (the reference
    detaches the plane images, and the arg-min input is built from them)."""
    h, w = src.shape[-2:]
    small = ops.resize_bilinear(target, cost_size)
    cand = ops.plane_sweep_depth_batch(K_cost, Kinv_cost, pose, depths, src, small, sid=sid)
    return ops.resize_bicubic(cand, (h, w))
