"""Input pipeline to the device: the last step of the reference's loaders, on the GPU.

`dataset.py:51,110` does `tf.to_tensor(tf.resize(img, (192, 640)))` per frame on the host and then `.to(device)`
(`dataset.py:55,113`) -- 12 B per pixel over PCIe.  Here the resized PIL frame travels as uint8 (3 B per pixel) and
`to_tensor` (uint8 HWC -> float32 CHW, value / 255) runs on the device.  `resize_to_tensor` takes the RAW frame
(1241x376 in KITTI: 1.4 MB instead of 1.5 MB of resized floats, and no host resize) and does PIL's antialiased bilinear
resize on the device too, bit for bit (Pillow's fixed-point arithmetic).
"""
from __future__ import annotations

import torch

from . import _lib as L


def to_tensor(frames: torch.Tensor) -> torch.Tensor:
    """frames (N,H,W,3) or (H,W,3) uint8 on the GPU -> (N,3,H,W) / (3,H,W) float32 in [0,1]; bit-identical to
    torchvision.transforms.functional.to_tensor on the same pixels."""
    if frames.dtype != torch.uint8 or not frames.is_cuda:
        raise RuntimeError(f"to_tensor expects a uint8 CUDA tensor, got {frames.dtype} on {frames.device}")
    single = frames.dim() == 3
    x = frames.unsqueeze(0) if single else frames
    if x.dim() != 4 or x.shape[-1] != 3:
        raise RuntimeError(f"expected (N,H,W,3) or (H,W,3), got {tuple(frames.shape)}")
    x = x.contiguous()
    n, h, w, _ = x.shape
    out = torch.empty(n, 3, h, w, dtype=torch.float32, device=x.device)
    L.check(L.lib().ds_ingest_u8(x.data_ptr(), n, h, w, out.data_ptr(), L.stream()), "ds_ingest_u8")
    return out[0] if single else out


def _resize(frames: torch.Tensor, size, want_u8: bool, want_f32: bool):
    if frames.dtype != torch.uint8 or not frames.is_cuda:
        raise RuntimeError(f"resize expects a uint8 CUDA tensor, got {frames.dtype} on {frames.device}")
    single = frames.dim() == 3
    x = frames.unsqueeze(0) if single else frames
    if x.dim() != 4 or x.shape[-1] != 3:
        raise RuntimeError(f"expected (N,H,W,3) or (H,W,3), got {tuple(frames.shape)}")
    x = x.contiguous()
    n, h, w, _ = x.shape
    ho, wo = int(size[0]), int(size[1])
    lib = L.lib()
    nb = lib.ds_ingest_resize_workspace_bytes(n, h, w, ho, wo)
    if nb == 0:
        raise RuntimeError(f"bad resize {h}x{w} -> {ho}x{wo}")
    ws = L.workspace.get(nb, x.device)
    u8 = torch.empty(n, ho, wo, 3, dtype=torch.uint8, device=x.device) if want_u8 else None
    f32 = torch.empty(n, 3, ho, wo, dtype=torch.float32, device=x.device) if want_f32 else None
    L.check(lib.ds_ingest_resize_u8(x.data_ptr(), n, h, w, ho, wo, L.ptr(u8), L.ptr(f32), ws.data_ptr(), nb, L.stream()),
            "ds_ingest_resize_u8")
    pick = lambda t: None if t is None else (t[0] if single else t)  # noqa: E731
    return pick(u8), pick(f32)


def resize_u8(frames: torch.Tensor, size) -> torch.Tensor:
    """tf.resize(img, size) of dataset.py:51,110 on the device: raw frames (N,H,W,3) or (H,W,3) uint8 on the GPU ->
    (N,Hout,Wout,3) uint8, bit-identical to PIL's Image.resize((Wout, Hout), BILINEAR); size = (Hout, Wout)."""
    return _resize(frames, size, True, False)[0]


def resize_to_tensor(frames: torch.Tensor, size) -> torch.Tensor:
    """tf.to_tensor(tf.resize(img, size)) (dataset.py:51,110) on the device: raw uint8 frames -> (N,3,Hout,Wout) float32."""
    return _resize(frames, size, False, True)[1]


def depth_to_tensor(depth: torch.Tensor) -> torch.Tensor:
    """Ground-truth depth (dataset.py:116-120): (N,H,W) or (H,W) 16-bit samples of the resized PNG on the GPU (uint16,
    or int16/int32 holding the same bit patterns / values as PIL modes I;16 and I give them) -> (N,1,H,W) / (1,H,W)
    float32 = value / 2**16."""
    if not depth.is_cuda:
        raise RuntimeError(f"depth_to_tensor expects a CUDA tensor, got {depth.device}")
    if depth.dtype == torch.int32:
        if depth.numel() and (int(depth.min()) < 0 or int(depth.max()) > 65535):
            raise RuntimeError("int32 depth samples must lie in [0, 65535]")
        depth = depth.to(torch.uint16)
    elif depth.dtype == torch.int16:
        depth = depth.view(torch.uint16)
    elif depth.dtype != torch.uint16:
        raise RuntimeError(f"depth_to_tensor expects uint16/int16/int32 samples, got {depth.dtype}")
    single = depth.dim() == 2
    x = (depth.unsqueeze(0) if single else depth).contiguous()
    if x.dim() != 3:
        raise RuntimeError(f"expected (N,H,W) or (H,W), got {tuple(depth.shape)}")
    n, h, w = x.shape
    out = torch.empty(n, 1, h, w, dtype=torch.float32, device=x.device)
    L.check(L.lib().ds_ingest_depth_u16(x.data_ptr(), n, h, w, out.data_ptr(), L.stream()), "ds_ingest_depth_u16")
    return out[0] if single else out
