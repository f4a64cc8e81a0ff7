"""Drop-in for the reference's metrics.py: same function names, argument order and semantics
(reference file:line in each docstring); every call runs HIP kernels on the tensors' GPU."""
from __future__ import annotations

import weakref

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


_shared_target_cache = None  # (weak reference to the last target stack examined, its version counter, the answer)


def _targets_are_one_image(targets: torch.Tensor) -> bool:
    """train.py:80 builds the target stack with torch.stack([target] * S): S identical copies.  An expanded view shows it
    in its strides; a real stack only in its values, which costs one comparison kernel and a host sync (the reference's
    loop syncs per sample anyway, train.py:92) -- remembered for the second call on the SAME tensor object, unmodified
    (train.py:84-85); another tensor, even one at the same address, is examined again."""
    global _shared_target_cache
    if targets.shape[0] == 1 or targets.stride(0) == 0:
        return True
    c = _shared_target_cache
    if c is not None and c[0]() is targets and c[1] == targets._version:
        return c[2]
    hit = bool((targets[1:] == targets[:1]).all().item())
    _shared_target_cache = (weakref.ref(targets), targets._version, hit)
    return hit


def reprojection_loss(preds: torch.Tensor, targets: torch.Tensor, shared_target: bool | None = None) -> torch.Tensor:
    """metrics.py:54-72 -- 0.85*SSIM + 0.15*L1, mean over channels, min over dim 0: (S,3,H,W) x2 -> (1,1,H,W).
    shared_target (not in the reference): True if every targets[s] is the same image (one launch for all sources instead
    of one per source; same result), None = find out."""
    if preds.dim() != 4 or targets.shape != preds.shape:
        raise RuntimeError(f"expected preds and targets (S,3,H,W); got {tuple(preds.shape)}, {tuple(targets.shape)}")
    s, c, h, w = preds.shape
    shared = _targets_are_one_image(targets) if shared_target is None else bool(shared_target)
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
    """metrics.py:114-125 (eps is fixed at the reference's default 1e-6 in the kernel)"""
    if eps != 1e-6:
        raise RuntimeError("rmsle: only the reference's default eps=1e-6 is supported")
    return ops.depth_metrics(pred_depths, depths)[1]


def abs_rel(pred_depths: torch.Tensor, depths: torch.Tensor) -> torch.Tensor:
    """metrics.py:128-139"""
    return ops.depth_metrics(pred_depths, depths)[2]


def sq_rel(pred_depths: torch.Tensor, depths: torch.Tensor) -> torch.Tensor:
    """metrics.py:142-153"""
    return ops.depth_metrics(pred_depths, depths)[3]
