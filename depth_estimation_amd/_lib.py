"""ctypes binding of libdepthsynth.so (include/depthsynth.h).

The product path has no fallback: if the HIP library is missing or a tensor is not a contiguous
float32 tensor on an AMD GPU, the call raises.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdepthsynth.so")

_fp = C.c_void_p
_i = C.c_int
_ll = C.c_longlong
_sz = C.c_size_t
_f32 = C.c_float

# name -> (restype, argtypes); mirrors include/depthsynth.h one to one
_SIGNATURES = {
    "ds_version": (_i, []),
    "ds_error_string": (C.c_char_p, [_i]),
    "ds_pose_from_vec_fwd": (_i, [_fp, _i, _fp, _fp]),
    "ds_pose_from_vec_bwd": (_i, [_fp, _fp, _i, _fp, _fp]),
    "ds_reproject": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp]),
    "ds_reproject_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp, _fp, _fp, _sz, _fp]),
    "ds_grid_sample_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp]),
    "ds_grid_sample_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp]),
    "ds_warp_fwd": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp]),
    "ds_warp_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "ds_warp_bwd_image_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ds_warp_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _sz, _fp]),
    "ds_synthesize_from_depths": (_i, [_fp, _fp, _fp, C.POINTER(C.c_double), _i, _fp, _i, _i, _i, _i, _i, _fp, _fp]),
    "ds_depth_from_costs": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _fp]),
    "ds_plane_sweep_depth": (_i, [_fp, _fp, _fp, C.POINTER(C.c_double), _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp,
                                  _fp]),
    "ds_plane_sweep_depth_batch": (_i, [_fp, _fp, _fp, C.POINTER(C.c_double), _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _i,
                                        _i, _fp, _fp]),
    "ds_ingest_depth_u16": (_i, [_fp, _i, _i, _i, _fp, _fp]),
    "ds_resize_bilinear": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _fp]),
    "ds_resize_bicubic": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _fp]),
    "ds_cost_volume": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp]),
    "ds_ssim_fwd": (_i, [_fp, _fp, _i, _i, _i, _fp, _fp]),
    "ds_ssim_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _fp, _fp]),
    "ds_reprojection_loss_fwd": (_i, [_fp, _fp, _ll, _ll, _i, _i, _i, _i, _fp, _fp, _fp]),
    "ds_reprojection_loss_bwd": (_i, [_fp, _fp, _ll, _ll, _fp, _fp, _i, _i, _i, _i, _fp, _fp]),
    "ds_fused_loss_workspace_bytes": (_sz, [_i, _i, _i]),
    "ds_fused_loss_fwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "ds_fused_loss_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp]),
    "ds_warp_identity_fwd": (_i, [_fp] * 6 + [_i] * 4 + [_fp, _fp, _fp]),
    "ds_fused_loss_fb_workspace_bytes": (_sz, [_i, _i, _i]),
    "ds_fused_loss_fb": (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "ds_fused_loss_fb_warp_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ds_fused_loss_fb_warp_bwd": (_i, [_fp] * 9 + [_i] * 4 + [_fp] * 6 + [_fp, _sz, _fp]),
    "ds_loss_step_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ds_loss_step_fwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "ds_loss_step_bwd": (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _sz, _fp]),
    "ds_loss_step_fwdbwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ds_loss_step_fwdbwd_image_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ds_loss_step_fwdbwd": (_i, [_fp] * 7 + [_i] * 4 + [_fp] * 9 + [_fp, _sz, _fp]),
    "ds_train_step_fwdbwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ds_train_step_fwdbwd": (_i, [_fp] * 6 + [_f32] + [_i] * 4 + [_fp] * 10 + [_fp, _sz, _fp]),
    "ds_smooth_workspace_bytes": (_sz, [_i, _i, _i]),
    "ds_smooth_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _fp, _fp, _sz, _fp]),
    "ds_smooth_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp]),
    "ds_smooth_batch_fwd": (_i, [_fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _sz, _fp]),
    "ds_smooth_batch_bwd": (_i, [_fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp]),
    "ds_ingest_u8": (_i, [_fp, _i, _i, _i, _fp, _fp]),
    "ds_ingest_resize_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ds_ingest_resize_u8": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _sz, _fp]),
    "ds_metrics_workspace_bytes": (_sz, [_i, _i, _i]),
    "ds_depth_metrics": (_i, [_fp, _fp, _i, _i, _i, _fp, _fp, _fp, _sz, _fp]),
    "ds_depth_metrics_eps": (_i, [_fp, _fp, _i, _i, _i, _f32, _fp, _fp, _fp, _sz, _fp]),
    "ds_depth_metrics_noalloc": (_i, [_fp, _fp, _i, _i, _i, _fp, _fp, _fp]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None


def lib() -> C.CDLL:
    """Load libdepthsynth.so (built by `make -C depth_estimation_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C depth_estimation_amd/csrc` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ds_error_string(int(rc))
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else 'unknown'}")


def stream() -> int:
    """Raw handle of torch's current stream on the current device.  torch.cuda.current_stream() builds a Python object
    per call (~12 us each: four of them were a tenth of a small loss step), so the raw lookup torch itself uses is
    preferred where this torch build has it; the public call is the fallback."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream


def dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    """Validate a tensor for the kernels: CUDA(HIP) device, float32; returns it contiguous."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU (got device {t.device}); "
                           "depth_estimation_amd has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def same_device(*ts: torch.Tensor) -> None:
    d = ts[0].device
    for t in ts[1:]:
        if t.device != d:
            raise RuntimeError(f"expected all tensors on {d}, got {t.device}")
    if d.index is not None and d.index != torch._C._cuda_getDevice():
        # the kernels are launched on the current device's stream
        raise RuntimeError(f"tensors are on {d} but the current device is cuda:{torch._C._cuda_getDevice()}; "
                           "call torch.cuda.set_device first")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


class Workspace:
    """Scratch buffer per (device, stream), reused across calls (kernels never allocate); calls on one stream are
    ordered, calls on different streams must not share partial sums.  A buffer only grows; `release()` drops them all
    (e.g. after a one-off call at a much larger size, or when streams come and go)."""

    def __init__(self):
        self._buf = {}

    def release(self) -> None:
        """Forget every buffer (torch frees them once the kernels already queued on their streams have run)."""
        self._buf.clear()

    def nbytes(self) -> int:
        return sum(b.numel() for b in self._buf.values())

    def get(self, nbytes: int, device: torch.device) -> torch.Tensor:
        key = (device.index, stream())
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf


workspace = Workspace()
