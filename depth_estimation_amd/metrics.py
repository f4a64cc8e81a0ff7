"""Drop-in for the reference's metrics.py: same function names, argument order and semantics
(reference file:line in each docstring); every call runs HIP kernels on the tensors' GPU."""
from __future__ import annotations

import torch

from . import ops


def cost_volumes(target_img: torch.Tensor, synthesized_imgs: torch.Tensor) -> torch.Tensor:
    """metrics.py:5-18 -- mean over channels of |target - synthesized|: (C,H,W) [or (1,C,H,W)], (N,C,H,W) -> (N,H,W)."""
    return ops.cost_volume(target_img, synthesized_imgs)


def reprojection_error(preds: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """metrics.py:21-51 -- clamped SSIM loss map with 3x3 mean filters over reflect-padded images."""
    if targets.dim() < 4:
        targets = targets.unsqueeze(0)
    if preds.dim() < 4:
        preds = preds.unsqueeze(0)
    return ops.ssim_loss(preds, targets)


def reprojection_loss(preds: torch.Tensor, targets: torch.Tensor, shared_target: bool | None = None) -> torch.Tensor:
    """metrics.py:54-72 -- 0.85*SSIM + 0.15*L1, mean over channels, min over dim 0: (S,3,H,W) x2 -> (1,1,H,W).

    Argument forms as the reference takes them: rank-3 inputs count as a batch of one (metrics.py:29-30; the L1 term
    broadcasts), and a dim 0 of size one broadcasts against the other argument's -- ONE target image (3,H,W) or (1,3,H,W)
    for S predictions, or one prediction against S targets.

    shared_target (not in the reference): True = the caller's promise that every targets[s] holds the same image, as
    train.py:80's torch.stack([target] * S) does; all S sources then go through one launch that forms the target's window
    sums once.  None (default): shared exactly when that is visible without reading the data -- one target image, or an
    expanded view (stride 0 along dim 0); a materialised stack takes one launch per source with a running minimum.  The
    result is the same bit for bit either way, and no form reads values back to the host.  Hot loops over train.py:84-85
    should pass shared_target=True (or call fused.train_step_loss, which never builds the stack)."""
    if preds.dim() == 3:
        preds = preds.unsqueeze(0)
    if targets.dim() == 3:
        targets = targets.unsqueeze(0)
    if preds.dim() != 4 or targets.dim() != 4 or preds.shape[1:] != targets.shape[1:] or \
            (preds.shape[0] != targets.shape[0] and 1 not in (preds.shape[0], targets.shape[0])):
        raise RuntimeError(f"expected preds and targets (S,3,H,W) [or (3,H,W)]; got {tuple(preds.shape)}, {tuple(targets.shape)}")
    if preds.shape[0] == 1 and targets.shape[0] > 1:
        preds = preds.expand(targets.shape[0], -1, -1, -1)    # one prediction against S targets (made contiguous below)
    s, c, h, w = preds.shape
    if shared_target is None:
        shared = targets.shape[0] == 1 or targets.stride(0) == 0
    else:
        shared = bool(shared_target)
    if not shared and targets.shape[0] != s:
        targets = targets.expand(s, -1, -1, -1)
    tg = targets[0].unsqueeze(0) if shared else targets.unsqueeze(0)
    out, _ = ops.reprojection_loss(preds.unsqueeze(0), tg, shared)
    return out.reshape(1, 1, h, w)


def regularization_term(depths: torch.Tensor, target_imgs: torch.Tensor) -> torch.Tensor:
    """metrics.py:75-97 -- edge-aware smoothness: mean of |d_x D| exp(-|d_x I|) + |d_y D| exp(-|d_y I|)."""
    return ops.smoothness(depths, target_imgs)


def depth_metrics(pred_depths: torch.Tensor, depths: torch.Tensor) -> dict:
    """All of metrics.py:100-153 in one pass, plus delta<1.25^k (not in the reference; same all-pixel
    denominator) and the valid-pixel count.  Returns 0-dim tensors (no host sync), plus "counts": (B,4) int64, per
    image the exact numbers of valid pixels with max(p/g, g/p) < 1.25, 1.25^2, 1.25^3 and of valid (gt > 0) pixels."""
    o, counts = ops.depth_metrics(pred_depths, depths, want_counts=True)
    names = ("rmse", "rmsle", "abs_rel", "sq_rel", "delta1", "delta2", "delta3", "n_valid")
    r = {k: o[i] for i, k in enumerate(names)}
    r["counts"] = counts
    return r


def rmse(pred_depths, depths: torch.Tensor) -> torch.Tensor:
    """metrics.py:100-111"""
    return ops.depth_metrics(pred_depths, depths)[0]


def rmsle(pred_depths: torch.Tensor, depths: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """metrics.py:114-125"""
    return ops.depth_metrics(pred_depths, depths, eps=eps)[1]


def abs_rel(pred_depths: torch.Tensor, depths: torch.Tensor) -> torch.Tensor:
    """metrics.py:128-139"""
    return ops.depth_metrics(pred_depths, depths)[2]


def sq_rel(pred_depths: torch.Tensor, depths: torch.Tensor) -> torch.Tensor:
    """metrics.py:142-153"""
    return ops.depth_metrics(pred_depths, depths)[3]
