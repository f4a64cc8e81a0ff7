"""depth_estimation_amd -- MI355X (gfx950) implementation of the view-synthesis / reprojection-loss hot
path of banaga-benjamin/Depth_Estimation.

    from depth_estimation_amd import synthesis, metrics     # the reference's module names and signatures
    from depth_estimation_amd import fused                  # batched, fused entry points
    from depth_estimation_amd import ingest                 # uint8 frames -> float tensors on the device

Everything runs in hand-written HIP kernels behind the C ABI of include/depthsynth.h
(depth_estimation_amd/libdepthsynth.so); there is no CPU fallback.
"""
from . import _lib  # noqa: F401
from . import fused, ingest, metrics, ops, synthesis  # noqa: F401

__all__ = ["synthesis", "metrics", "fused", "ingest", "ops"]
__version__ = "0.1.0"
