#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Build-container only: imports /root/reference/{synthesis,metrics,constants}.py (the
reference's own PyTorch code, CPU) and records inputs + per-stage outputs for the
hot path.  Nothing from the reference's sources is stored: a
fixture holds input tensors and the tensors the reference returned for them.

The reference cannot travel to the GPU box; the .npz files written here can and do.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

Call sequence reproduced per case (reference file:line):
    synthesis.construct_pose_batch          synthesis.py:6-33     (train.py:46)
    synthesis.reproject_from_depth          synthesis.py:83-116   (train.py:72-75)
    synthesis.synthesize                    synthesis.py:36-54    (train.py:77)
    metrics.reprojection_error / _loss      metrics.py:21-72      (train.py:84-85)
    ratio + regularization_term             train.py:83,88-89, metrics.py:75-97
    synthesis.synthesize_from_depths        synthesis.py:57-80    (train.py:49-52)
    metrics.cost_volumes, depth_from_costs  metrics.py:5-18, synthesis.py:119-138
    metrics.rmse/rmsle/abs_rel/sq_rel       metrics.py:100-153    (test.py:78-81)
"""
import os
import sys
import zlib

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import constants as ref_constants  # noqa: E402  (reference)
import metrics as ref_metrics  # noqa: E402  (reference)
import synthesis as ref_synthesis  # noqa: E402  (reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
torch.set_num_threads(4)


def crc(a: np.ndarray) -> np.ndarray:
    return np.array(zlib.crc32(np.ascontiguousarray(a).tobytes()), dtype=np.uint32)


def npf(t: torch.Tensor) -> np.ndarray:
    return np.array(t.detach().cpu().numpy(), copy=True, order="C")  # a copy: synthesize() mutates its grid in place


def intrinsics(width: int, height: int):
    """K rows scaled by (W, H) and pinv inverse, exactly as constants.py:33-44 does."""
    K = ref_constants.INTRINSIC_MAT.clone()
    K[0, :] *= width
    K[1, :] *= height
    return K, torch.linalg.pinv(K)


def load_frames(idx, width, height):
    """test_data_copy PNG -> PIL bilinear (W,H) -> uint8 HWC (dataset.py:45-51 on PIL input)."""
    out = []
    for i in idx:
        im = Image.open(f"{REF}/test_data_copy/{i:010d}.png").convert("RGB")
        im = im.resize((width, height), Image.BILINEAR)
        out.append(np.asarray(im, dtype=np.uint8))
    return np.stack(out)  # (F, H, W, 3) uint8


def frames_to_float(u8: np.ndarray) -> torch.Tensor:
    # to_tensor: uint8 HWC -> float32 CHW / 255
    return torch.from_numpy(u8).permute(0, 3, 1, 2).float().div(255)


def run_loss_case(name, src, target, depth, pose_vec, K, Kinv, full=True, sub=1):
    """One sample of train.py:46,72-92 + backward.  src (S,3,H,W), target (3,H,W),
    depth (S,1,H,W) in metres, pose_vec (S,6).  full=False stores strided samples + CRCs only."""
    S, C, H, W = src.shape
    rec = {}
    rec["src"] = npf(src)
    rec["target"] = npf(target)
    rec["depth"] = npf(depth)
    rec["pose_vec"] = npf(pose_vec)
    rec["K"] = npf(K)
    rec["Kinv"] = npf(Kinv)

    pose_vec = pose_vec.clone().requires_grad_(True)
    depth = depth.clone().requires_grad_(True)
    src_w = src.clone().requires_grad_(True)  # image-gradient path: warp only

    pose_mats = ref_synthesis.construct_pose_batch(pose_vec)
    pose_mats.retain_grad()
    coords = [ref_synthesis.reproject_from_depth(K, Kinv, pm, d, width=W, height=H)
              for pm, d in zip(pose_mats, depth)]
    proj = torch.stack(coords)
    coords_px = npf(proj)  # before synthesize() normalises it in place
    warped = ref_synthesis.synthesize(src_w, proj)
    warped.retain_grad()
    coords_norm = npf(proj)  # after the in-place normalisation (synthesis.py:50-51)

    target_imgs = torch.stack([target] * S, dim=0)
    reg = ref_metrics.regularization_term(depth / ref_constants.MAX_DEPTH, target_imgs)
    ssim_w = ref_metrics.reprojection_error(warped, target_imgs)
    out_l = ref_metrics.reprojection_loss(warped, target_imgs)
    idt_l = ref_metrics.reprojection_loss(src, target_imgs)
    ratio = out_l / torch.exp(idt_l + 1e-6)
    loss_photo = torch.mean(torch.mean(ratio, dim=(2, 3)))
    loss = loss_photo + 0.02 * reg
    log_val = torch.mean(torch.mean(out_l, dim=(1, 2, 3)))
    loss.backward()

    # per-source loss maps and argmin, recomputed through the same reference calls
    per_src = torch.stack([ref_metrics.reprojection_loss(warped[s:s + 1].detach(), target_imgs[s:s + 1])[0]
                           for s in range(S)])  # (S,1,H,W)
    argmin = per_src.min(dim=0)[1].to(torch.uint8)

    stages = {
        "pose_mats": npf(pose_mats), "coords_px": coords_px, "coords_norm": coords_norm,
        "warped": npf(warped), "ssim_warped": npf(ssim_w), "per_src_loss": npf(per_src),
        "out_loss": npf(out_l), "idt_loss": npf(idt_l), "argmin": npf(argmin), "ratio": npf(ratio),
        "g_depth": npf(depth.grad), "g_pose_mats": npf(pose_mats.grad), "g_pose_vec": npf(pose_vec.grad),
        "g_src": npf(src_w.grad), "g_warped": npf(warped.grad),
    }
    scalars = {"loss_photo": loss_photo, "reg": reg, "loss": loss, "log_val": log_val}
    for k, v in scalars.items():
        rec[k] = np.array(v.item(), dtype=np.float64)
    for k, v in stages.items():
        rec["crc_" + k] = crc(v)
        rec["sum_" + k] = np.array(v.astype(np.float64).sum())
        rec["abssum_" + k] = np.array(np.abs(v.astype(np.float64)).sum())
        if full or v.size <= 4096:
            rec[k] = v
        else:
            rec["sub_" + k] = np.ascontiguousarray(v[..., ::sub, :] if k.startswith("coords") else v[..., ::sub, ::sub])
    rec["sub"] = np.array(sub)

    # smoothness gradient alone (metrics.py:75-97) for the smoothness kernels
    d2 = torch.from_numpy(rec["depth"]).requires_grad_(True)
    reg2 = ref_metrics.regularization_term(d2 / ref_constants.MAX_DEPTH, target_imgs)
    reg2.backward()
    g = npf(d2.grad)
    rec["crc_g_depth_reg"] = crc(g)
    rec["sum_g_depth_reg"] = np.array(g.astype(np.float64).sum())
    if full:
        rec["g_depth_reg"] = g
    else:
        rec["sub_g_depth_reg"] = np.ascontiguousarray(g[..., ::sub, ::sub])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(f"{name}: loss={loss.item():.6f} photo={loss_photo.item():.6f} reg={reg.item():.6f} "
          f"argmin hist={np.bincount(stages['argmin'].ravel(), minlength=S)}")


def smooth_depth(S, H, W, gen, lo=5.0, hi=80.0):
    """ramp hi->lo down the image plus seeded low-amplitude noise."""
    ramp = torch.linspace(hi, lo, H).view(1, 1, H, 1).expand(S, 1, H, W)
    return (ramp + 0.5 * torch.rand(S, 1, H, W, generator=gen)).contiguous()


def case_real(name, width, height, full, sub):
    gen = torch.Generator().manual_seed(1234)
    u8 = load_frames([0, 1, 2, 3], width, height)
    imgs = frames_to_float(u8)
    src, target = imgs[:-1].contiguous(), imgs[-1].contiguous()
    K, Kinv = intrinsics(width, height)
    pose_vec = torch.tensor([[0.02, -0.03, 0.01, 0.3, -0.05, 0.8],
                             [-0.01, 0.015, 0.004, -0.2, 0.02, -0.5],
                             [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]]) + 0.01 * torch.randn(3, 6, generator=gen)
    depth = smooth_depth(3, height, width, gen)
    run_loss_case(name, src, target, depth, pose_vec, K, Kinv, full=full, sub=sub)
    np.savez_compressed(os.path.join(OUT, name + "_frames_u8.npz"), frames=u8)


def case_synth(name, S, H, W, seed, pose_scale, big_motion=False, ties=False, zcross=False, Hs=None, Ws=None):
    gen = torch.Generator().manual_seed(seed)
    Hs = Hs or H
    Ws = Ws or W
    src = torch.rand(S, 3, Hs, Ws, generator=gen)
    target = torch.rand(3, H, W, generator=gen)
    if ties:
        src[1:] = src[0]
    K, Kinv = intrinsics(W, H)
    pose_vec = pose_scale * torch.randn(S, 6, generator=gen)
    if ties:
        pose_vec[1:] = pose_vec[0]
    if big_motion:
        pose_vec[0] += torch.tensor([0.05, -0.2, 0.03, 2.0, -0.4, 1.5])
    depth = 1.0 + 79.0 * torch.rand(S, 1, H, W, generator=gen)
    if ties:
        depth[1:] = depth[0]
    if zcross:
        # tz ~ -depth puts projected z around 0 / negative for part of the image (synthesis.py:113)
        # zero rotation makes R = I exactly, so z = d + tz: exactly 0 where d = 40, negative where d < 40
        pose_vec[0, :3] = 0.0
        pose_vec[0, 5] = -40.0
        depth[0, 0, 0, :8] = 40.0
    if Hs != H or Ws != W:
        return src, target, depth, pose_vec, K, Kinv
    run_loss_case(name, src, target, depth, pose_vec, K, Kinv, full=True)


def case_srcdims(name):
    """source dims != grid dims: grid 24x80 in its own pixel units, source 48x160."""
    gen = torch.Generator().manual_seed(77)
    H, W, Hs, Ws = 24, 80, 48, 160
    src = torch.rand(2, 3, Hs, Ws, generator=gen)
    K, Kinv = intrinsics(W, H)
    pose_vec = 0.02 * torch.randn(2, 6, generator=gen)
    depth = 1.0 + 79.0 * torch.rand(2, 1, H, W, generator=gen)
    pose = ref_synthesis.construct_pose_batch(pose_vec)
    proj = torch.stack([ref_synthesis.reproject_from_depth(K, Kinv, pm, d, width=W, height=H)
                        for pm, d in zip(pose, depth)])
    coords_px = npf(proj)
    warped = ref_synthesis.synthesize(src, proj)
    # rank-3 inputs (synthesis.py:45-46)
    proj3 = torch.from_numpy(coords_px[0].copy())
    warped3 = ref_synthesis.synthesize(src[0], proj3)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), src=npf(src), depth=npf(depth), pose_vec=npf(pose_vec),
                        pose_mats=npf(pose), K=npf(K), Kinv=npf(Kinv), coords_px=coords_px, coords_norm=npf(proj),
                        warped=npf(warped), warped_rank3=npf(warped3), coords_norm_rank3=npf(proj3))
    print(name, "ok")


def case_plane_sweep(name):
    """train.py:49-60 at reduced size: src 48x160, cost grid 24x80, all 80 SID planes, S=2 sources."""
    gen = torch.Generator().manual_seed(99)
    Hs, Ws, H, W = 48, 160, 24, 80
    u8 = load_frames([10, 11, 12], Ws, Hs)
    imgs = frames_to_float(u8)
    src, target = imgs[:-1].contiguous(), imgs[-1].contiguous()
    K, Kinv = intrinsics(W, H)  # cost intrinsics are scaled by the COST dims (constants.py:33-37)
    pose_vec = torch.tensor([[0.004, -0.006, 0.002, 0.05, -0.01, 0.6],
                             [-0.003, 0.002, 0.001, -0.04, 0.01, -0.5]]) + 0.002 * torch.randn(2, 6, generator=gen)
    pose = ref_synthesis.construct_pose_batch(pose_vec)
    depths = ref_constants.DEPTHS
    synth = [ref_synthesis.synthesize_from_depths(K, Kinv, pm, depths, W, H, s) for pm, s in zip(pose, src)]
    tgt_small = torch.nn.functional.interpolate(target.unsqueeze(0), size=(H, W), mode="bilinear")
    cvs = torch.stack([ref_metrics.cost_volumes(tgt_small, s) for s in synth])
    cand_sid = ref_synthesis.depth_from_costs(cvs, num_channels=80, sid=True)
    cand_uid = ref_synthesis.depth_from_costs(cvs, num_channels=80, sid=False)
    syn0 = npf(synth[0])
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), src=npf(src), target=npf(target), target_small=npf(tgt_small),
        pose_vec=npf(pose_vec), pose_mats=npf(pose), K=npf(K), Kinv=npf(Kinv), depths=np.asarray(depths, dtype=np.float64),
        synth0_planes=np.ascontiguousarray(syn0[[0, 1, 17, 40, 79]]), synth0_plane_ids=np.array([0, 1, 17, 40, 79]),
        crc_synth0=crc(syn0), crc_synth1=crc(npf(synth[1])), cost_volumes=npf(cvs),
        cand_sid=npf(cand_sid), cand_uid=npf(cand_uid))
    print(name, "cand range", cand_sid.min().item(), cand_sid.max().item())


def case_metrics(name):
    """test.py:73-81: pred (B,1,H,W), GT with ~30% zeros."""
    gen = torch.Generator().manual_seed(5)
    B, H, W = 3, 40, 96
    gt = 1.0 + 79.0 * torch.rand(B, 1, H, W, generator=gen)
    gt[torch.rand(B, 1, H, W, generator=gen) < 0.3] = 0.0
    pred = (gt + 4.0 * torch.randn(B, 1, H, W, generator=gen)).clamp_min(0.05)
    pred[gt == 0] = 1.0 + 79.0 * torch.rand(int((gt == 0).sum()), generator=gen)
    rec = dict(pred=npf(pred), gt=npf(gt))
    for fn in ("rmse", "rmsle", "abs_rel", "sq_rel"):
        rec[fn] = np.array(getattr(ref_metrics, fn)(pred, gt).item(), dtype=np.float64)
    # all-invalid image and single image edge cases
    gt0 = torch.zeros(1, 1, H, W)
    for fn in ("rmse", "rmsle", "abs_rel", "sq_rel"):
        rec[fn + "_allzero"] = np.array(getattr(ref_metrics, fn)(pred[:1], gt0).item(), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, {k: float(v) for k, v in rec.items() if v.ndim == 0})


def case_pose(name):
    """construct_pose_batch incl. zero rotation and its gradient (synthesis.py:6-33)."""
    gen = torch.Generator().manual_seed(11)
    vec = torch.cat([0.01 * torch.randn(4, 6, generator=gen), torch.zeros(1, 6),
                     torch.tensor([[1.2, -0.7, 0.4, 3.0, -2.0, 1.0]])]).requires_grad_(True)
    w = torch.randn(6, 3, 4, generator=gen)
    pm = ref_synthesis.construct_pose_batch(vec)
    (pm * w).sum().backward()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), vec=npf(vec), pose_mats=npf(pm), w=npf(w), g_vec=npf(vec.grad))
    print(name, "ok")


def case_resize(name):
    """the two functional.interpolate calls of train.py:55 and :60, at the plane-sweep fixture's sizes and at ragged ones."""
    ps = np.load(os.path.join(OUT, "plane_sweep_24x80.npz"))
    cand = torch.from_numpy(ps["cand_sid"])                       # (2,1,24,80), train.py:59
    cand_full = torch.nn.functional.interpolate(cand, size=(48, 160), mode="bicubic")      # train.py:60
    gen = torch.Generator().manual_seed(123)
    a = torch.rand(2, 3, 37, 53, generator=gen)
    b = torch.rand(3, 1, 19, 20, generator=gen)
    rec = dict(cand_full=npf(cand_full), a=npf(a), b=npf(b))
    for tag, x, size in (("a_19x20", a, (19, 20)), ("a_50x131", a, (50, 131)), ("a_1x1", a, (1, 1)),
                         ("b_37x53", b, (37, 53)), ("b_7x64", b, (7, 64))):
        rec["bilinear_" + tag] = npf(torch.nn.functional.interpolate(x, size=size, mode="bilinear"))
        rec["bicubic_" + tag] = npf(torch.nn.functional.interpolate(x, size=size, mode="bicubic"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "ok")


LOSS_CASES = ["synth_24x80", "synth_oov_24x80", "synth_ties_24x80", "synth_zcross_24x80", "synth_odd_19x37", "real_48x160"]


def case_ssim_grad(name):
    """autograd of metrics.reprojection_error alone (metrics.py:21-51) w.r.t. preds under a seeded upstream gradient, on
    the warped images and targets the loss fixtures already hold.  Reads those fixtures, changes none of them: the
    forward map recomputed here must reproduce their stored CRC."""
    rec = {}
    for case in LOSS_CASES:
        d = np.load(os.path.join(OUT, case + ".npz"))
        warped = torch.from_numpy(d["warped"]).requires_grad_(True)
        S = warped.shape[0]
        target_imgs = torch.stack([torch.from_numpy(d["target"])] * S, dim=0)
        gen = torch.Generator().manual_seed(zlib.crc32(case.encode()) & 0xffff)
        g_up = torch.randn(warped.shape, generator=gen)
        ssim = ref_metrics.reprojection_error(warped, target_imgs)
        assert int(crc(npf(ssim))) == int(d["crc_ssim_warped"]), case       # existing fixtures are reproduced bit for bit
        ssim.backward(g_up)
        rec[case + "/g_up"] = npf(g_up)
        rec[case + "/g_ssim"] = npf(warped.grad)
        print(f"{name}: {case} |g|max = {warped.grad.abs().max().item():.4f}")
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)


def case_raw_frame(name):
    """One raw 1241x376 frame of test_data_copy as uint8 (data, not source): the input of the loaders' resize
    (dataset.py:44-51).  Its PIL-resized forms are frame 0 of real_192x640_frames_u8 / real_48x160_frames_u8, which this
    checks before writing."""
    im = Image.open(f"{REF}/test_data_copy/{0:010d}.png").convert("RGB")
    raw = np.asarray(im, dtype=np.uint8)
    for fx, (w, h) in (("real_192x640", (640, 192)), ("real_48x160", (160, 48))):
        have = np.load(os.path.join(OUT, fx + "_frames_u8.npz"))["frames"][0]
        assert np.array_equal(np.asarray(im.resize((w, h), Image.BILINEAR), dtype=np.uint8), have), fx
    np.savez_compressed(os.path.join(OUT, name + ".npz"), frame=raw)
    print(name, raw.shape)


def case_facade(name):
    """Round 3: argument forms of metrics.py the drop-in must accept like the reference does -- rmsle with a non-default
    eps (metrics.py:114), reprojection_loss on rank-3 inputs (metrics.py:29-30 unsqueeze them; the L1 term broadcasts)
    and on one target image against S predictions / one prediction against S targets (plain broadcasting)."""
    m = np.load(os.path.join(OUT, "metrics_3x40x96.npz"))
    pred, gt = torch.from_numpy(m["pred"]), torch.from_numpy(m["gt"])
    rec = {}
    for tag, eps in (("1e-3", 1e-3), ("0.5", 0.5), ("0", 0.0)):
        rec["rmsle_eps_" + tag] = np.array(ref_metrics.rmsle(pred, gt, eps=eps).item(), dtype=np.float64)
    d = np.load(os.path.join(OUT, "synth_oov_24x80.npz"))
    warped, target = torch.from_numpy(d["warped"]), torch.from_numpy(d["target"])      # (3,3,24,80), (3,24,80)
    S = warped.shape[0]
    rec["rank3_both"] = npf(ref_metrics.reprojection_loss(warped[0], target))           # (C,H,W) x2 -> (1,1,H,W)
    rec["rank3_target"] = npf(ref_metrics.reprojection_loss(warped, target))            # (S,C,H,W), (C,H,W)
    assert int(crc(rec["rank3_target"])) == int(d["crc_out_loss"])                      # == the stacked-target call
    tstack = torch.stack([target, warped[1], target.flip(-1)])                          # three different target images
    rec["tstack"] = npf(tstack)
    rec["rank3_pred"] = npf(ref_metrics.reprojection_loss(warped[0], tstack))           # (C,H,W), (S,C,H,W)
    rec["rank4_distinct_targets"] = npf(ref_metrics.reprojection_loss(warped, tstack))
    rec["err_rank3"] = npf(ref_metrics.reprojection_error(warped[0], target))           # (1,C,H,W)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, {k: (float(v) if v.ndim == 0 else v.shape) for k, v in rec.items()})


def case_plane_sweep_full(name):
    """train.py:49-60 at the reference's real shape: sources 192x640 (frames 0-2 of real_192x640_frames_u8), target frame
    3, cost grid COST_HEIGHT x COST_WIDTH = 96x320 with constants.COST_INTRINSIC_MAT / _INV, all 80 planes of
    constants.DEPTHS.  Stores the candidate depths (S,1,96,320) and every 4th row/column of their bicubic up-resize."""
    u8 = np.load(os.path.join(OUT, "real_192x640_frames_u8.npz"))["frames"]
    imgs = frames_to_float(u8)
    src, target = imgs[:-1].contiguous(), imgs[-1].contiguous()
    K, Kinv = ref_constants.COST_INTRINSIC_MAT, ref_constants.COST_INTRINSIC_INV
    H, W = ref_constants.COST_HEIGHT, ref_constants.COST_WIDTH
    K2, Kinv2 = intrinsics(W, H)
    assert torch.equal(K, K2) and torch.equal(Kinv, Kinv2)        # the restated intrinsics ARE the reference's
    pose_vec = torch.tensor([[0.004, -0.006, 0.002, 0.25, -0.01, 0.6],
                             [-0.003, 0.002, 0.001, -0.3, 0.01, -0.5],
                             [0.001, 0.001, -0.002, 0.1, 0.0, 1.0]])
    pose = ref_synthesis.construct_pose_batch(pose_vec)
    depths = ref_constants.DEPTHS
    synth = [ref_synthesis.synthesize_from_depths(K, Kinv, pm, depths, W, H, s) for pm, s in zip(pose, src)]
    tgt_small = torch.nn.functional.interpolate(target.unsqueeze(0), size=(H, W), mode="bilinear")
    cvs = torch.stack([ref_metrics.cost_volumes(tgt_small, s) for s in synth])
    cand = ref_synthesis.depth_from_costs(cvs)
    full = torch.nn.functional.interpolate(cand, size=(ref_constants.HEIGHT, ref_constants.WIDTH), mode="bicubic")
    np.savez_compressed(os.path.join(OUT, name + ".npz"), pose_vec=npf(pose_vec), pose_mats=npf(pose), K=npf(K), Kinv=npf(Kinv),
                        depths=np.asarray(depths, dtype=np.float64), cand=npf(cand), sub_full=npf(full[..., ::4, ::4]),
                        crc_synth=np.array([int(crc(npf(s))) for s in synth], dtype=np.uint32),
                        crc_cost=crc(npf(cvs)), sub_cost=npf(cvs[:, ::8, ::4, ::4]))
    print(name, "cand range", cand.min().item(), cand.max().item(), "std", cand.std().item())


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["facade"]:                                # round 3; reads two fixtures, rewrites none
        case_facade("facade_vectors")
        return
    if sys.argv[1:] == ["plane_sweep_full"]:                      # round 3; reads one fixture, rewrites none
        case_plane_sweep_full("plane_sweep_96x320")
        return
    if sys.argv[1:] == ["resize"]:                                # added later; leaves the other fixtures untouched
        case_resize("resize_vectors")
        return
    if sys.argv[1:] == ["raw_frame"]:                             # round 2; reads two fixtures, rewrites none
        case_raw_frame("raw_frame_376x1241_u8")
        return
    if sys.argv[1:] == ["ssim_grad"]:                             # round 2; reads the loss fixtures, rewrites none
        case_ssim_grad("ssim_grad")
        return
    case_real("real_48x160", 160, 48, full=True, sub=1)
    case_real("real_192x640", 640, 192, full=False, sub=8)       # BASELINE config 1
    case_synth("synth_24x80", S=2, H=24, W=80, seed=0, pose_scale=0.01)
    case_synth("synth_oov_24x80", S=3, H=24, W=80, seed=1, pose_scale=0.02, big_motion=True)
    case_synth("synth_ties_24x80", S=3, H=24, W=80, seed=2, pose_scale=0.01, ties=True)
    case_synth("synth_zcross_24x80", S=2, H=24, W=80, seed=3, pose_scale=0.01, zcross=True)
    case_synth("synth_odd_19x37", S=2, H=19, W=37, seed=4, pose_scale=0.02)  # ragged: not a tile multiple
    case_srcdims("srcdims_24x80_from_48x160")
    case_plane_sweep("plane_sweep_24x80")
    case_metrics("metrics_3x40x96")
    case_pose("pose_batch")
    case_resize("resize_vectors")


if __name__ == "__main__":
    main()
