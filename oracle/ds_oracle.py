"""numpy/ctypes front-end of the CPU parity oracle (oracle/ds_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  Parity status: pinned by tests/golden/*.npz
(outputs of the reference itself, see tools/gen_golden.py).

Every function takes/returns float32 numpy arrays in the reference's layouts (NCHW planar images,
(H,W,2) x-first coordinate grids).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libds_oracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    """Compile oracle/ds_oracle.c -> oracle/libds_oracle.so (gcc, see oracle/Makefile)."""
    src = os.path.join(_HERE, "ds_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libds_oracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.dso_depth_metrics.argtypes = [_f32p, _f32p, C.c_int, C.c_size_t, _f64p]
        _lib.dso_depth_metrics_eps.argtypes = [_f32p, _f32p, C.c_int, C.c_size_t, C.c_float, _f64p]
        _lib.dso_depth_metric_counts.argtypes = [_f32p, _f32p, C.c_int, C.c_size_t, C.POINTER(C.c_longlong)]
        _lib.dso_depth_from_costs.argtypes = [_f32p, C.c_int, C.c_int, C.c_size_t, C.c_int, _f32p]
        _lib.dso_regularization_term.restype = C.c_double
        _lib.dso_regularization_term.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _f32p]
        _lib.dso_loss_assembly_bwd.argtypes = [_f32p, _f32p, _f32p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_double, _f32p]
    return _lib


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def set_threads(n: int) -> None:
    lib().dso_set_threads(int(n))


def max_threads() -> int:
    return int(lib().dso_get_max_threads())


def pose_from_vec(vec) -> np.ndarray:
    vec = _f(vec).reshape(-1, 6)
    out = np.empty((vec.shape[0], 3, 4), np.float32)
    lib().dso_pose_from_vec(_p(vec), vec.shape[0], _p(out))
    return out


def reproject(K, Kinv, pose, depth, H: int, W: int) -> np.ndarray:
    """a2: one depth map -> (H,W,2) pixel coords."""
    K, Kinv, pose, depth = _f(K).reshape(4, 4), _f(Kinv).reshape(4, 4), _f(pose).reshape(3, 4), _f(depth).reshape(H, W)
    out = np.empty((H, W, 2), np.float32)
    lib().dso_reproject(_p(K), _p(Kinv), _p(pose), _p(depth), H, W, _p(out))
    return out


def grid_sample(src, grid_px):
    """a3 for one image: src (C,Hs,Ws), grid (H,W,2) pixel coords.
    Returns (out (C,H,W), grid_normalised (H,W,2)) -- the reference normalises its argument in place."""
    src = _f(src)
    Cc, Hs, Ws = src.shape
    grid = _f(grid_px).copy()
    H, W, _ = grid.shape
    out = np.empty((Cc, H, W), np.float32)
    lib().dso_grid_sample(_p(src), Cc, Hs, Ws, _p(grid), H, W, _p(out))
    return out, grid


def warp(K, Kinv, pose, depth, src, H: int, W: int) -> np.ndarray:
    """fused a2+a3 for one (source, depth, pose) triple -> warped (C,H,W)."""
    K, Kinv, pose, depth, src = _f(K), _f(Kinv), _f(pose), _f(depth).reshape(H, W), _f(src)
    Cc, Hs, Ws = src.shape
    out = np.empty((Cc, H, W), np.float32)
    lib().dso_warp(_p(K), _p(Kinv), _p(pose), _p(depth), _p(src), Cc, Hs, Ws, H, W, _p(out))
    return out


def ssim(preds, targets) -> np.ndarray:
    """a7: (N,C,H,W) x2 -> (N,C,H,W) clamped SSIM loss."""
    preds, targets = _f(preds), _f(targets)
    H, W = preds.shape[-2:]
    n = preds.size // (H * W)
    out = np.empty_like(preds)
    lib().dso_ssim(_p(preds), _p(targets), n, H, W, _p(out))
    return out


def reprojection_loss(preds, target):
    """a8: preds (S,C,H,W), ONE target (C,H,W) -> (out (H,W), argmin (H,W) u8, per_src (S,H,W))."""
    preds, target = _f(preds), _f(target)
    S, Cc, H, W = preds.shape
    out = np.empty((H, W), np.float32)
    am = np.empty((H, W), np.uint8)
    per = np.empty((S, H, W), np.float32)
    lib().dso_reprojection_loss(_p(preds), _p(target), S, Cc, H, W, _p(out), am.ctypes.data_as(_u8p), _p(per))
    return out, am, per


def loss_assembly(warped, src, target):
    """a9 for one target: returns dict(out_loss, idt_loss, ratio, argmin, mean_ratio, mean_out)."""
    warped, src, target = _f(warped), _f(src), _f(target)
    S, Cc, H, W = warped.shape
    out = np.empty((H, W), np.float32)
    idt = np.empty((H, W), np.float32)
    ratio = np.empty((H, W), np.float32)
    am = np.empty((H, W), np.uint8)
    sc = np.zeros(2, np.float64)
    lib().dso_loss_assembly(_p(warped), _p(src), _p(target), S, Cc, H, W, _p(out), _p(idt), _p(ratio),
                            am.ctypes.data_as(_u8p), sc.ctypes.data_as(_f64p))
    return dict(out_loss=out, idt_loss=idt, ratio=ratio, argmin=am, mean_ratio=float(sc[0]), mean_out=float(sc[1]))


def loss_assembly_bwd(warped, target, idt_loss, argmin, g_scale: float = 1.0) -> np.ndarray:
    warped, target, idt_loss = _f(warped), _f(target), _f(idt_loss)
    argmin = np.ascontiguousarray(argmin, dtype=np.uint8)
    S, Cc, H, W = warped.shape
    g = np.empty_like(warped)
    lib().dso_loss_assembly_bwd(_p(warped), _p(target), _p(idt_loss), argmin.ctypes.data_as(_u8p), S, Cc, H, W,
                                float(g_scale), _p(g))
    return g


def warp_bwd(K, Kinv, pose, depth, src, g_warped, H: int, W: int, want_src_grad: bool = True):
    """backward of one warp: returns (g_depth (H,W), g_pose (3,4), g_src (C,Hs,Ws) or None)."""
    K, Kinv, pose, depth, src, g_warped = _f(K), _f(Kinv), _f(pose), _f(depth).reshape(H, W), _f(src), _f(g_warped)
    Cc, Hs, Ws = src.shape
    gd = np.empty((H, W), np.float32)
    gp = np.empty((3, 4), np.float32)
    gs = np.zeros_like(src) if want_src_grad else None
    lib().dso_warp_bwd(_p(K), _p(Kinv), _p(pose), _p(depth), _p(src), _p(g_warped), Cc, Hs, Ws, H, W, _p(gd), _p(gp),
                       _p(gs) if gs is not None else None)
    return gd, gp, gs


def grid_sample_bwd(src, grid_px, g_out, want_src_grad: bool = True):
    src, grid_px, g_out = _f(src), _f(grid_px), _f(g_out)
    Cc, Hs, Ws = src.shape
    H, W, _ = grid_px.shape
    gg = np.empty((H, W, 2), np.float32)
    gs = np.zeros_like(src) if want_src_grad else None
    lib().dso_grid_sample_bwd(_p(src), Cc, Hs, Ws, _p(grid_px), H, W, _p(g_out), _p(gg),
                              _p(gs) if gs is not None else None)
    return gg, gs


def regularization_term(depths, targets, want_grad: bool = False, g_scale: float = 1.0):
    """a10: depths (N,1,H,W) (already /MAX_DEPTH), targets (N,C,H,W) -> scalar [, g_depths]."""
    depths, targets = _f(depths), _f(targets)
    N, Cc, H, W = targets.shape
    g = np.empty((N, 1, H, W), np.float32) if want_grad else None
    val = lib().dso_regularization_term(_p(depths), _p(targets), N, Cc, H, W, float(g_scale), _p(g) if want_grad else None)
    return (float(val), g) if want_grad else float(val)


def depth_metrics(pred, gt, eps: float = 1e-6) -> dict:
    """a11-a15: pred, gt (B,1,H,W); eps = the offset in rmsle's logarithms (metrics.py:114)."""
    pred, gt = _f(pred), _f(gt)
    B = pred.shape[0]
    hw = pred.size // B
    out = np.zeros(8, np.float64)
    lib().dso_depth_metrics_eps(_p(pred), _p(gt), B, hw, float(eps), out.ctypes.data_as(_f64p))
    keys = ("rmse", "rmsle", "abs_rel", "sq_rel", "delta1", "delta2", "delta3", "n_valid")
    return dict(zip(keys, out.tolist()))


def depth_metric_counts(pred, gt) -> np.ndarray:
    """a15 as integers: (B,4) int64 = per image #(delta<1.25), #(<1.25^2), #(<1.25^3), #(gt>0)."""
    pred, gt = _f(pred), _f(gt)
    B = pred.shape[0]
    out = np.zeros((B, 4), np.int64)
    lib().dso_depth_metric_counts(_p(pred), _p(gt), B, pred.size // B, out.ctypes.data_as(C.POINTER(C.c_longlong)))
    return out


def synthesize_from_depths(K, Kinv, pose, depths, W: int, H: int, src) -> np.ndarray:
    """a4: (D,C,H,W)."""
    K, Kinv, pose, src = _f(K), _f(Kinv), _f(pose), _f(src)
    depths = np.ascontiguousarray(depths, dtype=np.float64)
    Cc, Hs, Ws = src.shape
    out = np.empty((len(depths), Cc, H, W), np.float32)
    lib().dso_synthesize_from_depths(_p(K), _p(Kinv), _p(pose), depths.ctypes.data_as(_f64p), len(depths), _p(src), Cc,
                                     Hs, Ws, H, W, _p(out))
    return out


def cost_volume(target, synth) -> np.ndarray:
    """a5: target (C,H,W), synth (D,C,H,W) -> (D,H,W)."""
    target, synth = _f(target), _f(synth)
    D, Cc, H, W = synth.shape
    out = np.empty((D, H, W), np.float32)
    lib().dso_cost_volume(_p(target.reshape(Cc, H, W)), _p(synth), D, Cc, H, W, _p(out))
    return out


def depth_from_costs(cost, sid: bool = True) -> np.ndarray:
    """a6: cost (N,D,H,W) -> (N,1,H,W)."""
    cost = _f(cost)
    N, D, H, W = cost.shape
    out = np.empty((N, 1, H, W), np.float32)
    lib().dso_depth_from_costs(_p(cost), N, D, H * W, int(sid), _p(out))
    return out


def _resize(fn, x, size) -> np.ndarray:
    x = _f(x)
    Hin, Win = x.shape[-2:]
    out = np.empty(x.shape[:-2] + tuple(size), np.float32)
    fn(_p(x), int(np.prod(x.shape[:-2], dtype=np.int64)), Hin, Win, size[0], size[1], _p(out))
    return out


def resize_bilinear(x, size) -> np.ndarray:
    """functional.interpolate(x, size=size, mode="bilinear") of train.py:55; x (...,Hin,Win)."""
    return _resize(lib().dso_resize_bilinear, x, size)


def resize_bicubic(x, size) -> np.ndarray:
    """functional.interpolate(x, size=size, mode="bicubic") of train.py:60; x (...,Hin,Win)."""
    return _resize(lib().dso_resize_bicubic, x, size)


def pil_resize_bilinear_u8(frame, size) -> np.ndarray:
    """f.4: PIL Image.resize(size, BILINEAR) of a (H,W,C) uint8 frame (what dataset.py:51,110 does through tf.resize);
    size = (Hout, Wout) -> (Hout,Wout,C) uint8."""
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    H, W, Cc = frame.shape
    ho, wo = int(size[0]), int(size[1])
    tmp = np.empty((H, wo, Cc), np.uint8)
    out = np.empty((ho, wo, Cc), np.uint8)
    rc = lib().dso_pil_resize_bilinear_u8(frame.ctypes.data_as(_u8p), H, W, Cc, ho, wo, tmp.ctypes.data_as(_u8p),
                                          out.ctypes.data_as(_u8p))
    if rc != 0:
        raise ValueError("scale factor beyond the 64-tap limit")
    return out
