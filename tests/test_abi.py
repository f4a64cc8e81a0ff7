"""CPU-only: the C-ABI library loads and exports exactly what include/depthsynth.h declares."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "depthsynth.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from depth_estimation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libdepthsynth.so not built (run __graft_entry__.build())")
    handle = _lib.lib()
    declared = _declared()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in depthsynth.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "python binding table and header disagree"
    assert handle.ds_version() >= 100
    assert b"invalid argument" in handle.ds_error_string(-1)
    # argument validation happens on the host, before any launch: safe without a GPU
    assert handle.ds_warp_fwd(None, None, None, None, None, 1, 3, 8, 8, 8, 8, None, None) == -1
    assert handle.ds_fused_loss_workspace_bytes(12, 192, 640) > 0
    assert handle.ds_warp_bwd_workspace_bytes(24, 192, 640) == 24 * 240 * 12 * 4   # the short tiling of small launches


def test_every_entry_point_rejects_null_arguments_on_the_host():
    """Argument validation precedes any launch (error behaviour): every int-returning entry point called
    with null pointers and zero sizes returns an error code instead of touching the device -- safe without a GPU."""
    import ctypes as C
    from depth_estimation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libdepthsynth.so not built (run __graft_entry__.build())")
    handle = _lib.lib()
    checked = 0
    for name, (restype, argtypes) in _lib._SIGNATURES.items():
        if restype is not C.c_int or not argtypes:
            continue
        args = [None if a in (C.c_void_p, C.POINTER(C.c_double)) else 0 for a in argtypes]
        rc = getattr(handle, name)(*args)
        assert rc in (-1, -2, -3), f"{name}(nulls) returned {rc}"
        assert handle.ds_error_string(rc)
        checked += 1
    assert checked >= 25


def test_host_wrappers_refuse_cpu_tensors():
    import torch
    from depth_estimation_amd import metrics, synthesis
    with pytest.raises(RuntimeError):
        metrics.reprojection_loss(torch.zeros(2, 3, 8, 8), torch.zeros(2, 3, 8, 8))
    with pytest.raises(RuntimeError):
        synthesis.construct_pose_batch(torch.zeros(2, 6))
    with pytest.raises(RuntimeError):
        synthesis.synthesize(torch.zeros(1, 3, 8, 8), torch.zeros(1, 8, 8, 2))


def test_image_gradient_limits_and_workspace_errors_on_the_host():
    """The atomic-free image gradient of ds_warp_bwd / ds_loss_step_fwdbwd has limits of its own (a point entry packs row and
    column into 32 bits) and a workspace of its own: both are checked before any launch."""
    from depth_estimation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libdepthsynth.so not built (run __graft_entry__.build())")
    h = _lib.lib()
    fake = 0x1000                      # never dereferenced on the host
    N, H, W = 2, 24, 80
    small = h.ds_warp_bwd_workspace_bytes(N, H, W)
    big = h.ds_warp_bwd_image_workspace_bytes(N, 3, H, W, H, W)
    assert big > small > 0
    assert h.ds_warp_bwd_image_workspace_bytes(N, 4, H, W, H, W) == small          # other channel counts: the atomics path
    assert h.ds_loss_step_fwdbwd_image_workspace_bytes(1, 2, H, W) > h.ds_loss_step_fwdbwd_workspace_bytes(1, 2, H, W)

    def bwd(hs, ws, nbytes, g_src=fake, wsp=fake):
        return h.ds_warp_bwd(fake, fake, fake, fake, fake, fake, N, 3, hs, ws, H, W, fake, fake, g_src, wsp, nbytes, None)

    assert bwd(H, W, small) == -3                      # DS_ERR_WORKSPACE: the staging does not fit the plain workspace
    assert bwd(H, W, big, wsp=fake + 4) == -3          # ... and must be 16-byte aligned
    assert bwd(40000, 8, 1 << 40) == -2                # DS_ERR_UNSUPPORTED: source rows beyond 32767
    assert bwd(8, 70000, 1 << 40) == -2                # source columns beyond 65534
    assert bwd(H, W, 0, g_src=None) == -3              # without the image gradient: the plain workspace is still required
    assert b"workspace" in h.ds_error_string(-3)
