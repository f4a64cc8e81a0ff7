"""Drop-in for the reference's synthesis.py: same function names, argument order and semantics
(reference file:line in each docstring); every call runs HIP kernels on the tensors' GPU.

The `device` arguments are kept for signature compatibility; work happens on the device the tensors
live on, and CPU tensors are rejected (no CPU fallback).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import ops


def construct_pose_batch(pose_vectors: torch.Tensor, device: str = "cpu") -> torch.Tensor:
    """synthesis.py:6-33 -- (N,6) axis-angle + translation -> (N,3,4) pose matrices (Rodrigues with the
    reference's theta+1e-6 normalisation).  Differentiable."""
    return ops.pose_from_vec(pose_vectors)


def synthesize(src_img: torch.Tensor, proj_pixels: torch.Tensor) -> torch.Tensor:
    """synthesis.py:36-54 -- bilinear, border-clamped, align_corners sampling of src_img (N,C,Hs,Ws) at
    proj_pixels (N,H,W,2) given in pixel units.  As in the reference, proj_pixels is normalised IN PLACE
    with the source dims (:49-51) and rank-3 inputs are treated as a batch of one (:45-46)."""
    if src_img.dim() < 4:
        src_img = src_img.unsqueeze(0)
    if proj_pixels.dim() < 4:
        proj_pixels = proj_pixels.unsqueeze(0)  # a view: the caller's tensor still sees the normalisation
    src = L.dev_f32(src_img, "src_img")
    if not (proj_pixels.is_cuda and proj_pixels.dtype == torch.float32):
        raise RuntimeError("proj_pixels must be a float32 tensor on the GPU")
    L.same_device(src, proj_pixels)
    if proj_pixels.shape[0] != src.shape[0] or proj_pixels.shape[-1] != 2:
        raise RuntimeError(f"expected src (N,C,H,W) and proj_pixels (N,H,W,2); got {tuple(src.shape)}, "
                           f"{tuple(proj_pixels.shape)}")
    if proj_pixels.is_contiguous():
        return ops.grid_sample_(src, proj_pixels)
    # non-contiguous grid: sample a packed copy, then write the normalised coordinates back
    packed = proj_pixels.contiguous()
    out = ops.grid_sample_(src, packed)
    with torch.no_grad():
        proj_pixels.copy_(packed)
    return out


def reproject_from_depth(intrinsic_mat: torch.Tensor, intrinsic_inv: torch.Tensor, pose_mat: torch.Tensor,
                         depth_map: torch.Tensor, width: int = 640, height: int = 192,
                         device: str = "cpu") -> torch.Tensor:
    """synthesis.py:83-116 -- back-project the (height,width) pixel grid with depth_map, apply pose_mat (3,4),
    project with intrinsic_mat -> (H,W,2) pixel coordinates, x first.  depth_map may be (1,H,W), (H,W) or
    (1,1,H,W) (:103-104).  Differentiable w.r.t. depth_map and pose_mat."""
    if depth_map.numel() != height * width:
        raise RuntimeError(f"depth_map has {depth_map.numel()} elements, expected {height}x{width}")
    coords = ops.reproject(intrinsic_mat, intrinsic_inv, pose_mat.reshape(1, 3, 4), depth_map.reshape(1, height, width))
    return coords[0]


def synthesize_from_depths(intrinsic_mat: torch.Tensor, intrinsic_inv: torch.Tensor, pose_mat: torch.Tensor,
                           depths: np.ndarray, width: int, height: int, src_img: torch.Tensor,
                           device: str = "cpu") -> torch.Tensor:
    """synthesis.py:57-80 -- plane sweep: one constant-depth warp of src_img (C,Hs,Ws) per entry of `depths`
    -> (D,C,height,width).  Forward only (every caller detaches it or runs under no_grad: train.py:51)."""
    return ops.synthesize_from_depths(intrinsic_mat, intrinsic_inv, pose_mat, depths, width, height, src_img)


def depth_from_costs(cost_volumes: torch.Tensor, num_channels: int = 80, sid: bool = True,
                     device: str = "cpu") -> torch.Tensor:
    """synthesis.py:119-138 -- softmin over the plane axis of exp(c)^2-1, expected plane index, SID/UID
    index->depth map.  (N,D,H,W) -> (N,1,H,W).  Forward only."""
    if cost_volumes.dim() != 4 or cost_volumes.shape[1] != num_channels:
        raise RuntimeError(f"cost_volumes must be (N,{num_channels},H,W), got {tuple(cost_volumes.shape)}")
    return ops.depth_from_costs(cost_volumes, sid)
