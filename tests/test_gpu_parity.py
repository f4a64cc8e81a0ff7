"""GPU parity: the HIP kernels (through the C ABI, via depth_estimation_amd) against the CPU oracle and the
golden vectors of the reference.  Run on the GPU box: pytest -m gpu.

Bars (BASELINE.json north_star): bit-exact for every forward stage without a transcendental (stricter than
the stated 1e-5); 1e-5 max-abs on anything that goes through expf; 1e-4 relative-to-max on gradients and
scalar metrics.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import LOSS_CASES, load_golden, real_frames, rel_to_max

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def de():
    import depth_estimation_amd as pkg
    pkg._lib.lib()
    assert torch.cuda.is_available()
    return pkg


def _case(name):
    d = load_golden(name)
    src, tgt, depth = d["src"], d["target"], d["depth"]
    S, C, H, W = src.shape
    return d, src, tgt, depth, d["K"], d["Kinv"], d["pose_mats"], S, C, H, W


@pytest.mark.parametrize("name", LOSS_CASES)
def test_warp_forward_bit_exact(de, oracle, name):
    d, src, tgt, depth, K, Kinv, pm, S, C, H, W = _case(name)
    # batched reproject (synthesis.py:83-116)
    coords = de.ops.reproject(T(K), T(Kinv), T(pm), T(depth.reshape(S, H, W)))
    assert np.array_equal(N(coords), d["coords_px"])
    # sampler with in-place normalisation (synthesis.py:36-54)
    grid = coords.clone()
    warped = de.ops.grid_sample_(T(src), grid)
    assert np.array_equal(N(grid), d["coords_norm"])
    assert np.array_equal(N(warped), d["warped"])
    # fused warp
    fused = de.ops.warp(T(K), T(Kinv), T(pm), T(depth.reshape(S, H, W)), T(src))
    assert np.array_equal(N(fused), d["warped"])
    ow = np.stack([oracle.warp(K, Kinv, pm[s], depth[s], src[s], H, W) for s in range(S)])
    assert np.array_equal(N(fused), ow)


@pytest.mark.parametrize("name", LOSS_CASES)
def test_photometric_forward(de, oracle, name):
    d, src, tgt, depth, K, Kinv, pm, S, C, H, W = _case(name)
    warped = d["warped"]
    tg = np.stack([tgt] * S)
    ssim = de.ops.ssim_loss(T(warped), T(tg))
    assert np.array_equal(N(ssim), d["ssim_warped"])                       # metrics.py:21-51, bit-exact
    out, aux = de.ops.reprojection_loss(T(warped[None]), T(tgt[None]), True)
    assert np.array_equal(N(out)[0], d["out_loss"].reshape(H, W))          # metrics.py:54-72, bit-exact
    assert np.array_equal(N(aux)[0] & 0x1f, d["argmin"].reshape(H, W))
    out2, aux2 = de.ops.reprojection_loss(T(warped[None]), T(tg[None]), False)  # per-source targets path
    assert np.array_equal(N(out2), N(out)) and np.array_equal(N(aux2) & 0x1f, N(aux) & 0x1f)
    sums, out3, idt, aux3 = de.ops.fused_loss(T(warped[None]), T(src[None]), T(tgt[None]))
    assert np.array_equal(N(out3), N(out))
    assert np.array_equal(N(idt)[0], d["idt_loss"].reshape(H, W))
    assert np.array_equal(N(aux3), N(aux))
    la = oracle.loss_assembly(warped, src, tgt)
    assert abs(float(sums[0, 0]) - la["mean_ratio"]) <= 1e-5 and abs(float(sums[0, 0]) - float(d["loss_photo"])) <= 1e-5
    assert abs(float(sums[0, 1]) - la["mean_out"]) <= 1e-5 and abs(float(sums[0, 1]) - float(d["log_val"])) <= 1e-5


@pytest.mark.parametrize("name", LOSS_CASES)
def test_photometric_backward(de, oracle, name):
    d, src, tgt, depth, K, Kinv, pm, S, C, H, W = _case(name)
    warped = T(d["warped"][None]).requires_grad_(True)
    sums, out, idt, aux = de.ops.fused_loss(warped, T(src[None]), T(tgt[None]))
    sums[:, 0].sum().backward()
    gw = N(warped.grad)[0]
    ogw = oracle.loss_assembly_bwd(d["warped"], tgt, d["idt_loss"].reshape(H, W), d["argmin"].reshape(H, W))
    assert rel_to_max(gw, ogw) < 1e-4
    assert rel_to_max(gw, d["g_warped"]) < 1e-4
    # reprojection_loss alone with an arbitrary upstream gradient
    g_up = np.random.default_rng(0).standard_normal((1, H, W)).astype(np.float32)
    p2 = T(d["warped"][None]).requires_grad_(True)
    o2, _ = de.ops.reprojection_loss(p2, T(tgt[None]), True)
    o2.backward(T(g_up))
    p3 = T(d["warped"][None]).requires_grad_(True)
    o3, _ = de.ops.reprojection_loss(p3, T(np.stack([tgt] * S)[None]), False)
    o3.backward(T(g_up))
    assert rel_to_max(N(p3.grad), N(p2.grad)) < 1e-6
    # consistency with the fused form: g_up = 1/(HW)/exp(idt+1e-6)
    g_eq = (1.0 / (H * W) / np.exp(d["idt_loss"].reshape(1, H, W).astype(np.float64) + 1e-6)).astype(np.float32)
    p4 = T(d["warped"][None]).requires_grad_(True)
    o4, _ = de.ops.reprojection_loss(p4, T(tgt[None]), True)
    o4.backward(T(g_eq))
    assert rel_to_max(N(p4.grad)[0], ogw) < 1e-4


@pytest.mark.parametrize("name", LOSS_CASES)
def test_ssim_backward(de, oracle, name):
    """autograd of metrics.reprojection_error alone (metrics.py:21-51) under a seeded upstream gradient, against the
    reference's own autograd (tests/golden/ssim_grad.npz, tools/gen_golden.py ssim_grad): 1e-4 of the largest gradient."""
    d, src, tgt, depth, K, Kinv, pm, S, C, H, W = _case(name)
    gold = load_golden("ssim_grad")
    x = T(d["warped"]).requires_grad_(True)
    ss = de.ops.ssim_loss(x, T(np.stack([tgt] * S)))
    assert np.array_equal(N(ss), d["ssim_warped"])
    ss.backward(T(gold[name + "/g_up"]))
    assert rel_to_max(N(x.grad), gold[name + "/g_ssim"]) < 1e-4
    # and the SSIM-only kernel against the fused one on a single source, where min/argmin are trivial
    x1 = T(d["warped"][:1]).requires_grad_(True)
    (de.ops.ssim_loss(x1, T(tgt[None])) * (0.85 / 3)).sum().backward()
    y = T(d["warped"][None, :1]).requires_grad_(True)
    o, _ = de.ops.reprojection_loss(y, T(tgt[None]), True)
    o.sum().backward()
    l1 = (0.15 / 3) * -np.sign(tgt[None] - d["warped"][:1])     # remove the L1 part
    assert rel_to_max(N(x1.grad), N(y.grad)[0] - l1) < 1e-5


@pytest.mark.parametrize("name", LOSS_CASES)
def test_warp_backward(de, oracle, name):
    d, src, tgt, depth, K, Kinv, pm, S, C, H, W = _case(name)
    dep = T(depth.reshape(S, H, W)).requires_grad_(True)
    pose = T(pm).requires_grad_(True)
    sr = T(src).requires_grad_(True)
    warped = de.ops.warp(T(K), T(Kinv), pose, dep, sr)
    warped.backward(T(d["g_warped"]))
    ogd, ogp, ogs = zip(*[oracle.warp_bwd(K, Kinv, pm[s], depth[s], src[s], d["g_warped"][s], H, W) for s in range(S)])
    assert rel_to_max(N(dep.grad), np.stack(ogd)) < 1e-4
    assert rel_to_max(N(pose.grad), np.stack(ogp)) < 1e-4
    assert rel_to_max(N(sr.grad), np.stack(ogs)) < 1e-4
    assert rel_to_max(N(pose.grad), d["g_pose_mats"]) < 1e-4
    assert rel_to_max(N(sr.grad), d["g_src"]) < 1e-4
    # unfused path: reproject -> synthesize, same gradients
    dep2 = T(depth.reshape(S, H, W)).requires_grad_(True)
    pose2 = T(pm).requires_grad_(True)
    coords = de.ops.reproject(T(K), T(Kinv), pose2, dep2)
    w2 = de.ops.grid_sample_(T(src), coords)
    w2.backward(T(d["g_warped"]))
    assert rel_to_max(N(dep2.grad), N(dep.grad)) < 1e-5
    assert rel_to_max(N(pose2.grad), N(pose.grad)) < 1e-5


@pytest.mark.parametrize("N_,C,H,W", [(1, 2, 20, 70), (2, 1, 37, 131), (3, 4, 9, 64)])
def test_warp_backward_other_channel_counts(de, oracle, N_, C, H, W):
    """ds_warp_bwd for C != 3 (the generic kernel; launches small enough that the 3-channel kernels would switch to their
    short tiles): depth, pose and image gradients against oracle.warp_bwd, 1e-4 of the largest."""
    rng = np.random.default_rng(C * 100 + H)
    K, Kinv = [a.numpy() for a in de.fused.intrinsics(W, H)]
    src = rng.random((N_, C, H, W), dtype=np.float32)
    depth = (1 + 79 * rng.random((N_, H, W), dtype=np.float32)).astype(np.float32)
    g_warped = rng.standard_normal((N_, C, H, W)).astype(np.float32)
    pm = oracle.pose_from_vec((0.01 * rng.standard_normal((N_, 6))).astype(np.float32))
    dep, pose, sr = T(depth).requires_grad_(True), T(pm).requires_grad_(True), T(src).requires_grad_(True)
    w = de.ops.warp(T(K), T(Kinv), pose, dep, sr)
    assert np.array_equal(N(w), np.stack([oracle.warp(K, Kinv, pm[n], depth[n], src[n], H, W) for n in range(N_)]))
    w.backward(T(g_warped))
    gd, gp, gs = zip(*[oracle.warp_bwd(K, Kinv, pm[n], depth[n], src[n], g_warped[n], H, W) for n in range(N_)])
    assert rel_to_max(N(dep.grad), np.stack(gd)) < 1e-4
    assert rel_to_max(N(pose.grad), np.stack(gp)) < 1e-4
    assert rel_to_max(N(sr.grad), np.stack(gs)) < 1e-4


@pytest.mark.parametrize("motion", ["subpixel", "disparity", "far"])
def test_image_gradient_any_motion(de, oracle, motion):
    """The image gradient of ds_warp_bwd collects a wave's taps in a register window that follows the warp and stages the
    rest as points: the result must not depend on where the taps land.  Sub-pixel motion (all in the window), depth-
    dependent disparity of tens of pixels (mixed), everything shifted out of the image (every pixel a point, all of them in
    the border rows: the rows' overflow lists) -- against the oracle, and bit-identical between two runs."""
    S, H, W = 2, 37, 131
    rng = np.random.default_rng(17)
    K, Kinv = [a.numpy() for a in de.fused.intrinsics(W, H)]
    src = rng.random((S, 3, H, W), dtype=np.float32)
    g_warped = rng.standard_normal((S, 3, H, W)).astype(np.float32)
    depth = {"subpixel": 1 + 79 * rng.random((S, H, W), dtype=np.float32),
             "disparity": 1 + 9 * rng.random((S, H, W), dtype=np.float32) ** 3,
             "far": 1 + 79 * rng.random((S, H, W), dtype=np.float32)}[motion].astype(np.float32)
    vec = {"subpixel": 0.01 * rng.standard_normal((S, 6)),
           "disparity": np.array([[0.01, -0.02, 0.005, 0.6, -0.05, 0.2], [-0.01, 0.015, 0.0, -0.5, 0.04, -0.1]]),
           "far": np.array([[0.3, 0.9, 0.1, 5.0, 1.0, 0.0], [0.0, -0.8, 0.2, -40.0, 3.0, 1.0]])}[motion].astype(np.float32)
    pm = oracle.pose_from_vec(vec)
    got = []
    for _ in range(2):
        sr = T(src).requires_grad_(True)
        de.ops.warp(T(K), T(Kinv), T(pm), T(depth), sr).backward(T(g_warped))
        got.append(N(sr.grad))
    want = np.stack([oracle.warp_bwd(K, Kinv, pm[s], depth[s], src[s], g_warped[s], H, W)[2] for s in range(S)])
    assert np.abs(want).max() > 0
    assert rel_to_max(got[0], want) < 1e-4
    assert np.array_equal(got[0], got[1])


def test_image_gradient_fuzz(de, oracle):
    """Forty seeded cases of the atomic-free image gradient against the oracle's adjoint: odd sizes (widths that are not
    multiples of 64, two-row images), source dims different from the target's, one to three images, pose scales from
    sub-pixel to everything out of view, random / smooth / constant depth (constant far planes pile taps onto single
    texels: the run sums), sparse upstream gradients.  Every case twice: equal bits."""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for case in range(40):
        n = int(rng.integers(1, 4))
        H, W = int(rng.integers(2, 70)), int(rng.integers(2, 200))
        Hs, Ws = (H, W) if case % 3 else (int(rng.integers(2, 70)), int(rng.integers(2, 200)))
        K, Kinv = [a.numpy() for a in de.fused.intrinsics(W, H)]
        src = rng.random((n, 3, Hs, Ws), dtype=np.float32)
        g_warped = rng.standard_normal((n, 3, H, W)).astype(np.float32)
        if case % 5 == 4:
            g_warped *= rng.random((n, 1, H, W)) < 0.1          # mostly zero upstream
        kind = case % 4
        depth = (1 + 79 * rng.random((n, H, W), dtype=np.float32) if kind == 0 else
                 np.broadcast_to(np.linspace(60, 3, H, dtype=np.float32)[None, :, None], (n, H, W)) + rng.random((n, H, W), dtype=np.float32) if kind == 1 else
                 np.full((n, H, W), 1e6 if kind == 2 else 2.5, dtype=np.float32))
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        scale = [0.003, 0.03, 0.3, 1.5][int(rng.integers(0, 4))]
        vec = (scale * rng.standard_normal((n, 6)) * np.array([1, 1, 1, 10, 3, 10])).astype(np.float32)
        pm = oracle.pose_from_vec(vec)
        got = []
        for _ in range(2):
            sr, dep, pose = T(src).requires_grad_(True), T(depth).requires_grad_(True), T(pm).requires_grad_(True)
            de.ops.warp(T(K), T(Kinv), pose, dep, sr).backward(T(g_warped))
            got.append((N(sr.grad), N(dep.grad), N(pose.grad)))
        assert all(np.array_equal(a, b) for a, b in zip(*got)), f"case {case}: two runs differ"
        gd, gp, gs = zip(*[oracle.warp_bwd(K, Kinv, pm[i], depth[i], src[i], g_warped[i], H, W) for i in range(n)])
        want = np.stack(gs)
        if np.abs(want).max() > 0:
            e = rel_to_max(got[0][0], want)
            worst = max(worst, e)
            assert e < 1e-4, f"case {case}: H={H} W={W} Hs={Hs} Ws={Ws} n={n} kind={kind} scale={scale}: {e}"
        else:
            assert not got[0][0].any()
    assert worst > 0


@pytest.mark.parametrize("name", LOSS_CASES)
def test_reference_call_sequence_end_to_end(de, name):
    """train.py:72-92 written against the drop-in modules, gradients through autograd, vs the reference's
    own numbers for the same inputs."""
    syn, met = de.synthesis, de.metrics
    d, src, tgt, depth, K, Kinv, pm, S, C, H, W = _case(name)
    Kt, Kit = T(K), T(Kinv)
    pose_mats = T(pm).requires_grad_(True)
    depth_outputs = T(depth).requires_grad_(True)       # (S,1,H,W) metres
    src_imgs, target_img = T(src), T(tgt)
    proj_pixels = torch.stack([syn.reproject_from_depth(Kt, Kit, p, dm, width=W, height=H, device=DEV)
                               for p, dm in zip(pose_mats, depth_outputs)])
    output_imgs = syn.synthesize(src_imgs, proj_pixels)
    assert np.array_equal(N(proj_pixels), d["coords_norm"])            # mutated in place like the reference
    assert np.array_equal(N(output_imgs), d["warped"])
    target_imgs = torch.stack([target_img] * S, dim=0)
    reg = met.regularization_term(depth_outputs / 80, target_imgs)
    output_reprojection = met.reprojection_loss(output_imgs, target_imgs)
    src_reprojection = met.reprojection_loss(src_imgs, target_imgs)
    assert output_reprojection.shape == (1, 1, H, W)
    assert np.array_equal(N(output_reprojection), d["out_loss"])
    assert np.array_equal(N(src_reprojection), d["idt_loss"])
    loss = torch.mean(torch.mean(output_reprojection / torch.exp(src_reprojection + 1e-6), dim=(2, 3)))
    loss = loss + 0.02 * reg
    assert abs(float(reg) - float(d["reg"])) < 1e-4 * abs(float(d["reg"]))
    assert abs(float(loss) - float(d["loss"])) < 1e-5
    loss.backward()
    assert rel_to_max(N(depth_outputs.grad), d["g_depth"]) < 1e-4
    assert rel_to_max(N(pose_mats.grad), d["g_pose_mats"]) < 1e-4


def test_fused_batched_matches_per_sample(de, oracle):
    """B targets in one launch == the same targets one at a time (what batch sharding relies on), and the
    fused entry point == the reference numbers, on BASELINE config 1 (192x640 frames, S=3)."""
    d = load_golden("real_192x640")
    fr = real_frames("real_192x640")
    src, tgt = fr[:-1], fr[-1]
    S, C, H, W = src.shape
    sub = int(d["sub"])
    K, Kinv = T(d["K"]), T(d["Kinv"])
    pose = T(d["pose_mats"]).requires_grad_(True)
    depth = T(d["depth"]).requires_grad_(True)
    srct = T(src).requires_grad_(True)
    r = de.fused.fused_reprojection_loss(srct[None], T(tgt)[None], depth[None], pose[None], K, Kinv, image_grad=True)
    assert np.array_equal(N(r["warped"])[0][..., ::sub, ::sub], d["sub_warped"])
    assert np.array_equal(N(r["out_loss"])[0][::sub, ::sub], d["sub_out_loss"].reshape(H // sub, W // sub))
    assert np.array_equal(N(r["idt_loss"])[0][::sub, ::sub], d["sub_idt_loss"].reshape(H // sub, W // sub))
    assert abs(float(r["loss"][0]) - float(d["loss_photo"])) < 1e-5
    assert abs(float(r["log"][0]) - float(d["log_val"])) < 1e-5
    reg = de.metrics.regularization_term(depth / 80, T(np.stack([tgt] * S)))
    (r["loss"].sum() + 0.02 * reg).backward()
    assert rel_to_max(N(depth.grad)[..., ::sub, ::sub], d["sub_g_depth"]) < 1e-4
    assert rel_to_max(N(pose.grad), d["g_pose_mats"]) < 1e-4
    assert rel_to_max(N(srct.grad)[..., ::sub, ::sub], d["sub_g_src"]) < 1e-4
    # batch of 3 permuted copies: every sample must reproduce the single-sample result bit for bit
    perm = [[0, 1, 2], [2, 0, 1], [1, 2, 0]]
    bs = torch.stack([T(src[p]) for p in perm])
    bd = torch.stack([T(d["depth"][p]) for p in perm])
    bp = torch.stack([T(d["pose_mats"][p]) for p in perm])
    rb = de.fused.fused_reprojection_loss(bs, torch.stack([T(tgt)] * 3), bd, bp, K, Kinv)
    for i in range(3):
        assert np.array_equal(N(rb["out_loss"])[i], N(r["out_loss"])[0])
        assert np.array_equal(N(rb["idt_loss"])[i], N(r["idt_loss"])[0])
    assert np.array_equal(N(rb["aux"])[0] & 0x1f, N(r["aux"])[0] & 0x1f)
    # per-workgroup fp32 partials: the row chunks, and with them the grouping of the sum, follow the batch size
    assert abs(float(rb["loss"][0]) - float(r["loss"][0])) < 1e-6 * abs(float(r["loss"][0]))


def test_source_dims_differ_from_grid_dims(de):
    d = load_golden("srcdims_24x80_from_48x160")
    w = de.ops.warp(T(d["K"]), T(d["Kinv"]), T(d["pose_mats"]), T(d["depth"].reshape(2, 24, 80)), T(d["src"]))
    assert np.array_equal(N(w), d["warped"])
    grid3 = T(d["coords_px"][0].copy())                      # rank-3 inputs (synthesis.py:45-46)
    out = de.synthesis.synthesize(T(d["src"][0]), grid3)
    assert np.array_equal(N(out), d["warped_rank3"])
    assert np.array_equal(N(grid3), d["coords_norm_rank3"])  # the caller's tensor was normalised in place
    # non-contiguous grid
    big = torch.zeros(24, 80, 4, device=DEV)
    big[..., :2] = T(d["coords_px"][0])
    view = big[..., :2]
    out2 = de.synthesis.synthesize(T(d["src"][0]), view)
    assert np.array_equal(N(out2), d["warped_rank3"]) and np.array_equal(N(view), d["coords_norm_rank3"])


def test_pose_batch(de):
    d = load_golden("pose_batch")
    v = T(d["vec"]).requires_grad_(True)
    pm = de.synthesis.construct_pose_batch(v, DEV)
    assert np.abs(N(pm) - d["pose_mats"]).max() < 1e-6
    (pm * T(d["w"])).sum().backward()
    assert rel_to_max(N(v.grad), d["g_vec"]) < 1e-4


def test_plane_sweep(de, oracle):
    d = load_golden("plane_sweep_24x80")
    K, Kinv, pm, src = T(d["K"]), T(d["Kinv"]), T(d["pose_mats"]), T(d["src"])
    syn0 = de.synthesis.synthesize_from_depths(K, Kinv, pm[0], d["depths"], 80, 24, src[0], DEV)
    assert np.array_equal(N(syn0)[d["synth0_plane_ids"]], d["synth0_planes"])             # synthesis.py:57-80
    assert np.array_equal(N(syn0), oracle.synthesize_from_depths(d["K"], d["Kinv"], d["pose_mats"][0], d["depths"], 80, 24, d["src"][0]))
    tsm = T(d["target_small"])
    cvs = torch.stack([de.metrics.cost_volumes(tsm, de.synthesis.synthesize_from_depths(K, Kinv, pm[s], d["depths"], 80, 24, src[s], DEV))
                       for s in range(2)])
    assert np.array_equal(N(cvs), d["cost_volumes"])                                      # metrics.py:5-18
    for sid, key in ((True, "cand_sid"), (False, "cand_uid")):
        cand = de.synthesis.depth_from_costs(cvs, num_channels=80, sid=sid, device=DEV)
        assert np.abs(N(cand) - d[key]).max() < 1e-5                                      # synthesis.py:119-138
        fused = de.ops.plane_sweep_depth(K, Kinv, pm, d["depths"], src, tsm, sid=sid)
        assert np.abs(N(fused) - d[key]).max() < 1e-5


def test_resizes(de, oracle):
    """train.py:55 / :60: against the torch-made vectors (1e-5, north_star) and the oracle (same operation order: 1e-6)."""
    d = load_golden("resize_vectors")
    ps = load_golden("plane_sweep_24x80")
    assert np.abs(N(de.ops.resize_bilinear(T(ps["target"]), (24, 80))) - ps["target_small"][0]).max() < 1e-5
    assert np.abs(N(de.ops.resize_bicubic(T(ps["cand_sid"]), (48, 160))) - d["cand_full"]).max() < 1e-5
    for key, tag, size in (("a", "a_19x20", (19, 20)), ("a", "a_50x131", (50, 131)), ("a", "a_1x1", (1, 1)),
                           ("b", "b_37x53", (37, 53)), ("b", "b_7x64", (7, 64))):
        for mode in ("bilinear", "bicubic"):
            got = N(getattr(de.ops, "resize_" + mode)(T(d[key]), size))
            assert got.shape == d[mode + "_" + tag].shape
            assert np.abs(got - d[mode + "_" + tag]).max() < 1e-5, (mode, tag)
            assert np.abs(got - getattr(oracle, "resize_" + mode)(d[key], size)).max() < 1e-6, (mode, tag)
    with pytest.raises(RuntimeError):
        de.ops.resize_bilinear(T(d["a"]), (0, 4))


def test_candidate_depths_batch(de, oracle):
    """train.py:49-60 for a batch in three launches == the per-target calls; golden: plane_sweep_24x80 + resize_vectors."""
    d = load_golden("plane_sweep_24x80")
    r = load_golden("resize_vectors")
    K, Kinv = T(d["K"]), T(d["Kinv"])
    # batch of 3 targets: the fixture's sequence, the same with sources swapped, and a synthetic one
    rng = np.random.default_rng(5)
    src = np.stack([d["src"], d["src"][::-1], rng.random(d["src"].shape, dtype=np.float32)])
    tgt = np.stack([d["target"], d["target"], rng.random(d["target"].shape, dtype=np.float32)])
    pose = np.stack([d["pose_mats"], d["pose_mats"][::-1], d["pose_mats"]])
    small = de.ops.resize_bilinear(T(tgt), (24, 80))
    got = de.ops.plane_sweep_depth_batch(K, Kinv, T(pose), d["depths"], T(src), small)
    assert got.shape == (3, 2, 1, 24, 80)
    for b in range(3):
        one = de.ops.plane_sweep_depth(K, Kinv, T(pose[b]), d["depths"], T(src[b]), small[b])
        assert np.array_equal(N(got[b]), N(one)), b
    assert np.abs(N(got[0]) - d["cand_sid"]).max() < 1e-5
    full = de.fused.candidate_depths(T(src), T(tgt), T(pose), K, Kinv, d["depths"], (24, 80))
    assert full.shape == (3, 2, 1, 48, 160)
    assert np.abs(N(full[0]) - r["cand_full"]).max() < 2e-5
    assert np.abs(N(full[1])[::-1] - r["cand_full"]).max() < 2e-5


def test_depth_metrics(de, oracle):
    d = load_golden("metrics_3x40x96")
    pred, gt = T(d["pred"]), T(d["gt"])
    for fn in ("rmse", "rmsle", "abs_rel", "sq_rel"):
        v = float(getattr(de.metrics, fn)(pred, gt))
        assert abs(v - float(d[fn])) < 1e-4 * abs(float(d[fn])), fn                       # metrics.py:100-153
        z = float(getattr(de.metrics, fn)(pred[:1], torch.zeros_like(gt[:1])))
        assert z == 0.0
    m = de.metrics.depth_metrics(pred, gt)
    o = oracle.depth_metrics(d["pred"], d["gt"])
    for k in ("delta1", "delta2", "delta3", "n_valid"):
        assert abs(float(m[k]) - o[k]) < 1e-4 * max(abs(o[k]), 1e-12), k


def test_smoothness(de, oracle):
    for name in ("real_48x160", "synth_odd_19x37"):
        d = load_golden(name)
        S, C, H, W = d["src"].shape
        dep = T(d["depth"]).requires_grad_(True)
        reg = de.metrics.regularization_term(dep / 80, T(np.stack([d["target"]] * S)))
        assert abs(float(reg) - float(d["reg"])) < 1e-4 * abs(float(d["reg"]))            # metrics.py:75-97
        reg.backward()
        assert rel_to_max(N(dep.grad), d["g_depth_reg"]) < 1e-4


@pytest.mark.parametrize("B,S,C,H,W", [(2, 1, 3, 37, 131), (3, 2, 3, 40, 72), (1, 5, 3, 33, 63), (2, 3, 3, 2, 2),
                                       (1, 2, 3, 70, 125), (2, 2, 1, 19, 37), (1, 1, 3, 17, 1)])
def test_smoothness_batch_forms(de, oracle, B, S, C, H, W):
    """metrics.regularization_term with the target stack implicit (targets (B,C,H,W) for B*S maps) == the stacked call ==
    the oracle (metrics.py:75-97), value and depth gradient to 1e-4; sizes around the 64/62-column and 8/16-row tiles,
    more maps per target than one pass takes, a channel count that takes the generic kernels, a one-column image."""
    rng = np.random.default_rng(B * 1000 + S * 100 + H)
    dep = (rng.random((B * S, 1, H, W), dtype=np.float32) + 0.01).astype(np.float32)
    tgt = rng.random((B, C, H, W), dtype=np.float32)
    stacked = np.repeat(tgt, S, axis=0)
    if H > 1 and W > 1:
        want, g_want = oracle.regularization_term(dep, stacked, want_grad=True)
    d1 = T(dep).requires_grad_(True)
    v1 = de.ops.smoothness(d1, T(tgt))
    v1.backward()
    d2 = T(dep).requires_grad_(True)
    v2 = de.ops.smoothness(d2, T(stacked))
    v2.backward()
    assert abs(float(v1) - float(v2)) <= 1e-6 * abs(float(v2))
    assert rel_to_max(N(d1.grad), N(d2.grad)) < 1e-6 or np.abs(N(d2.grad)).max() == 0
    if H > 1 and W > 1:
        assert abs(float(v1) - want) < 1e-4 * abs(want)
        assert rel_to_max(N(d1.grad), g_want) < 1e-4


def test_train_step_loss_matches_per_sample_composition(de):
    """fused.train_step_loss (train.py:83-89 for B targets at once, one autograd node) == the per-sample composition the
    reference writes: sum_b [ mean(out/exp(idt+1e-6)) + 0.02 * regularization_term(depth_b/80, target x S) ], values,
    depth and pose-vector gradients (1e-4 relative), also under a non-unit upstream gradient."""
    g = torch.Generator().manual_seed(7)
    B, S, H, W = 3, 2, 40, 72
    src = torch.rand(B, S, 3, H, W, generator=g).to(DEV)
    tgt = torch.rand(B, 3, H, W, generator=g).to(DEV)
    depth = (1 + 79 * torch.rand(B, S, 1, H, W, generator=g)).to(DEV)
    vec = (0.01 * torch.randn(B * S, 6, generator=g)).to(DEV)
    K, Kinv = [t.to(DEV) for t in de.fused.intrinsics(W, H)]
    d1, v1 = depth.clone().requires_grad_(True), vec.clone().requires_grad_(True)
    r = de.fused.train_step_loss(src, tgt, d1, de.ops.pose_from_vec(v1).reshape(B, S, 3, 4), K, Kinv)
    (r["total"] * 1.7).backward()
    d2, v2 = depth.clone().requires_grad_(True), vec.clone().requires_grad_(True)
    pose2 = de.ops.pose_from_vec(v2).reshape(B, S, 3, 4)
    total = 0.0
    for b in range(B):
        rb = de.fused.fused_reprojection_loss(src[b:b + 1], tgt[b:b + 1], d2[b:b + 1], pose2[b:b + 1], K, Kinv)
        total = total + rb["loss"].sum() + 0.02 * de.metrics.regularization_term(d2[b] / 80.0, torch.stack([tgt[b]] * S))
    (total * 1.7).backward()
    assert abs(float(r["total"]) - float(total)) < 1e-5 * abs(float(total))
    assert rel_to_max(N(d1.grad), N(d2.grad)) < 1e-4
    assert rel_to_max(N(v1.grad), N(v2.grad)) < 1e-4
    # the extras are those of fused_reprojection_loss
    rb = de.fused.fused_reprojection_loss(src, tgt, depth, pose2.detach(), K, Kinv)
    # (maps bit for bit; the per-target means come from another order of the partial sums in the fused step: 1e-6)
    assert np.allclose(N(r["loss"]), N(rb["loss"]), rtol=1e-6, atol=0) and np.array_equal(N(r["aux"]), N(rb["aux"]))
    assert np.array_equal(N(r["out_loss"]), N(rb["out_loss"])) and np.array_equal(N(r["idt_loss"]), N(rb["idt_loss"]))
    assert np.array_equal(N(r["out_loss"]), N(rb["out_loss"])) and np.array_equal(N(r["warped"]), N(rb["warped"]))


@pytest.mark.parametrize("name", LOSS_CASES + ["real_192x640"])
def test_train_step_loss_against_reference(de, name):
    """fused.train_step_loss (one autograd node, train.py:72-89) through construct_pose_batch against the reference's own
    numbers for the same inputs: loss (1e-5), d loss / d depth and d loss / d pose VECTORS (1e-4 of the largest)."""
    d = load_golden(name)
    if name == "real_192x640":
        fr = real_frames(name)
        src, tgt, sub = fr[:-1], fr[-1], int(d["sub"])
    else:
        src, tgt, sub = d["src"], d["target"], 1
    S, C, H, W = src.shape
    depth = T(d["depth"]).requires_grad_(True)                 # (S,1,H,W) metres
    vec = T(d["pose_vec"]).requires_grad_(True)
    pose = de.synthesis.construct_pose_batch(vec, DEV).reshape(1, S, 3, 4)
    r = de.fused.train_step_loss(T(src)[None], T(tgt)[None], depth[None], pose, T(d["K"]), T(d["Kinv"]),
                                 eager_grads=(S == 2))      # both forms of the step over the cases
    assert abs(float(r["total"]) - float(d["loss"])) < 1e-5
    assert abs(float(r["loss"][0]) - float(d["loss_photo"])) < 1e-5
    assert abs(float(r["reg"]) * 1.0 - float(d["reg"]) * 80.0) < 1e-4 * abs(float(d["reg"]) * 80.0)   # reg(d) = 80 reg(d/80)
    r["total"].backward()
    g_depth = d["sub_g_depth"] if "sub_g_depth" in d.files else d["g_depth"]
    assert rel_to_max(N(depth.grad)[..., ::sub, ::sub], g_depth) < 1e-4
    assert rel_to_max(N(vec.grad), d["g_pose_vec"]) < 1e-4


@pytest.mark.parametrize("B,S,Hs,Ws,H,W,depth_kind", [(12, 2, 192, 640, 192, 640, "random"), (12, 2, 192, 640, 192, 640, "smooth"),
                                                       (3, 2, 96, 320, 96, 320, "random"), (2, 2, 192, 640, 96, 320, "random"),
                                                       (2, 2, 48, 160, 96, 320, "smooth")])
def test_image_gradient_bench_case(de, oracle, B, S, Hs, Ws, H, W, depth_kind):
    """The image-gradient scatter of ds_warp_bwd on bench.py's own case (B=12, 192x640, random per-pixel depth: rows flip
    between the register window, take-over and direct scatter), at 96x320 and with source dims != grid dims, against
    oracle.warp_bwd (1e-4 of the largest).  Round 3: no float atomics -- every texel is a 64-bit fixed-point sum of the staged
    window rows and points, independent of the order in which they arrived: two runs must agree bit for bit."""
    g = torch.Generator().manual_seed(B * 1000 + H)
    N_ = B * S
    src = torch.rand(N_, 3, Hs, Ws, generator=g)
    if depth_kind == "random":
        depth = 1 + 79 * torch.rand(N_, H, W, generator=g)
    else:
        depth = torch.linspace(80, 5, H).view(1, H, 1).expand(N_, H, W) + 0.5 * torch.rand(N_, H, W, generator=g)
    depth = depth.contiguous()
    vec = 0.01 * torch.randn(N_, 6, generator=g)
    g_warped = torch.randn(N_, 3, H, W, generator=g)
    K, Kinv = de.fused.intrinsics(W, H)
    pose_np = oracle.pose_from_vec(vec.numpy())

    def run():
        sr = src.to(DEV).requires_grad_(True)
        de.ops.warp(K.to(DEV), Kinv.to(DEV), T(pose_np), depth.to(DEV), sr).backward(g_warped.to(DEV))
        return N(sr.grad)

    a, b = run(), run()
    assert np.abs(a).max() > 0 and np.array_equal(a, b)
    for n in range(N_):
        want = oracle.warp_bwd(K.numpy(), Kinv.numpy(), pose_np[n], depth[n].numpy(), src[n].numpy(), g_warped[n].numpy(),
                               H, W)[2]
        assert rel_to_max(a[n], want) < 1e-4, n


def test_delta_counts_are_exact_integers(de, oracle):
    """a15 (delta < 1.25^k; not in the reference, parity unpinned): the threshold counts are integer work, so the kernel's
    per-image counts must equal the oracle's and a direct numpy count exactly, with the reference's valid-pixel rule
    gt > 0 (metrics.py:108) and its all-pixel denominator for the means (metrics.py:111)."""
    d = load_golden("metrics_3x40x96")
    g = torch.Generator().manual_seed(9)
    big_gt = 1 + 79 * torch.rand(5, 1, 192, 640, generator=g)
    big_gt[torch.rand(big_gt.shape, generator=g) < 0.3] = 0.0
    big_pred = (big_gt + 6 * torch.randn(big_gt.shape, generator=g)).clamp_min(0.05)
    for pred, gt in ((d["pred"], d["gt"]), (big_pred.numpy(), big_gt.numpy())):
        m = de.metrics.depth_metrics(T(pred), T(gt))
        got = N(m["counts"])
        B = pred.shape[0]
        hw = pred[0].size
        assert got.dtype == np.int64 and got.shape == (B, 4)
        assert np.array_equal(got, oracle.depth_metric_counts(pred, gt))
        p, q = pred.reshape(B, -1).astype(np.float32), gt.reshape(B, -1).astype(np.float32)
        valid = q > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.maximum(p / q, q / p)
        want = np.stack([(valid & (r < np.float32(t))).sum(1) for t in (1.25, 1.5625, 1.953125)] + [valid.sum(1)], axis=1)
        assert np.array_equal(got, want)
        # the float outputs are those counts over the all-pixel denominator, averaged over the batch
        for k, name in enumerate(("delta1", "delta2", "delta3")):
            assert abs(float(m[name]) - (want[:, k] / hw).mean()) < 1e-6
        assert float(m["n_valid"]) == float(want[:, 3].sum())


@pytest.mark.parametrize("B,S,H,W", [(12, 2, 192, 640), (2, 2, 37, 131), (3, 1, 40, 72), (1, 2, 2, 2), (2, 2, 17, 63),
                                     (1, 2, 5, 3), (2, 1, 9, 70), (1, 2, 3, 129), (2, 2, 64, 62), (1, 2, 33, 125),
                                     (2, 3, 37, 131), (8, 3, 192, 640), (1, 3, 5, 70)])
def test_fused_forward_backward_kernel(de, B, S, H, W):
    """ds_fused_loss_fb (forward and backward of the warped groups in one kernel, identity map given) against the
    kernels it replaces, through the C ABI: out_loss and the argmin bit for bit, the means to 1e-6, d loss / d warped bit
    for bit (same arithmetic in the same order) -- with a non-unit upstream gradient per target; sizes at, below and
    off the 62-column band and the 2-row slab; S = 3 is refused by this entry point (DS_ERR_UNSUPPORTED; the variant with the
    warp's autograd takes it)."""
    L = de._lib
    lib, st = L.lib(), L.stream()
    g = torch.Generator().manual_seed(B * 1000 + H * 7 + W)
    warped = torch.rand(B, S, 3, H, W, generator=g).to(DEV)
    src = torch.rand(B, S, 3, H, W, generator=g).to(DEV)
    tgt = torch.rand(B, 3, H, W, generator=g).to(DEV)
    if B > 1 and S > 1:
        warped[1, 1] = warped[1, 0]        # a tie: the first index must win in both forms
    g_sums = (0.5 + torch.rand(B, generator=g)).to(DEV)
    f32 = dict(dtype=torch.float32, device=DEV)
    out1, idt, out2 = torch.empty(B, H, W, **f32), torch.empty(B, H, W, **f32), torch.empty(B, H, W, **f32)
    aux1 = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
    aux2 = torch.full((B, H, W), 77, dtype=torch.uint8, device=DEV)
    sums1, sums2 = torch.empty(B, 2, **f32), torch.empty(B, 2, **f32)
    gw1, gw2 = torch.empty_like(warped), torch.full_like(warped, float("nan"))
    n1 = lib.ds_fused_loss_workspace_bytes(B, H, W)
    ws1 = torch.empty(n1, dtype=torch.uint8, device=DEV)
    L.check(lib.ds_fused_loss_fwd(warped.data_ptr(), src.data_ptr(), tgt.data_ptr(), B, S, H, W, out1.data_ptr(), idt.data_ptr(),
                                  aux1.data_ptr(), sums1.data_ptr(), ws1.data_ptr(), n1, st), "ds_fused_loss_fwd")
    L.check(lib.ds_fused_loss_bwd(warped.data_ptr(), tgt.data_ptr(), idt.data_ptr(), aux1.data_ptr(), g_sums.data_ptr(), B, S,
                                  H, W, gw1.data_ptr(), st), "ds_fused_loss_bwd")
    # the identity map alone must be the fused kernel's
    idt2 = torch.empty(B, H, W, **f32)
    L.check(lib.ds_reprojection_loss_fwd(src.data_ptr(), tgt.data_ptr(), 3 * H * W, 0, B, S, H, W, idt2.data_ptr(), 0, st),
            "ds_reprojection_loss_fwd")
    assert torch.equal(idt2, idt)
    n2 = lib.ds_fused_loss_fb_workspace_bytes(B, H, W)
    ws2 = torch.empty(n2, dtype=torch.uint8, device=DEV)
    if S <= 2:
        L.check(lib.ds_fused_loss_fb(warped.data_ptr(), tgt.data_ptr(), idt2.data_ptr(), g_sums.data_ptr(), B, S, H, W,
                                     out2.data_ptr(), aux2.data_ptr(), sums2.data_ptr(), gw2.data_ptr(), ws2.data_ptr(), n2, st),
                "ds_fused_loss_fb")
        torch.cuda.synchronize()
        assert torch.equal(out2, out1)
        assert torch.equal(aux2 & 0x1f, aux1 & 0x1f)
        assert float(((sums2 - sums1).abs() / sums1.abs()).max()) < 1e-6
        assert torch.equal(gw2, gw1), float((gw2 - gw1).abs().max())
        if B > 1 and S > 1:
            assert not bool(((aux2[1] & 0x1f) == 1).any())
    # the same kernel with the warp's autograd in its fourth wave (ds_fused_loss_fb_warp_bwd; also S = 3): d loss / d warped
    # (optional output) and d loss / d depth bit for bit those of ds_fused_loss_bwd + ds_warp_bwd, the pose gradient to 1e-4
    depth = (1 + 79 * torch.rand(B, S, H, W, generator=g)).to(DEV)
    pose = de.ops.pose_from_vec((0.01 * torch.randn(B * S, 6, generator=g)).to(DEV)).contiguous()
    K, Kinv = [t.to(DEV) for t in de.fused.intrinsics(W, H)]
    gd1, gp1 = torch.empty(B, S, H, W, **f32), torch.empty(B * S, 3, 4, **f32)
    nw = lib.ds_warp_bwd_workspace_bytes(B * S, H, W)
    wsw = torch.empty(nw, dtype=torch.uint8, device=DEV)
    L.check(lib.ds_warp_bwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(), gw1.data_ptr(),
                            B * S, 3, H, W, H, W, gd1.data_ptr(), gp1.data_ptr(), 0, wsw.data_ptr(), nw, st), "ds_warp_bwd")
    n3 = lib.ds_fused_loss_fb_warp_bwd_workspace_bytes(B, S, H, W)
    ws3 = torch.empty(n3, dtype=torch.uint8, device=DEV)
    for want_gw in (True, False):
        out3, aux3 = torch.empty(B, H, W, **f32), torch.full((B, H, W), 99, dtype=torch.uint8, device=DEV)
        sums3, gw3 = torch.empty(B, 2, **f32), torch.full_like(warped, float("nan"))
        gd3, gp3 = torch.full((B, S, H, W), float("nan"), **f32), torch.full((B * S, 3, 4), float("nan"), **f32)
        L.check(lib.ds_fused_loss_fb_warp_bwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                              warped.data_ptr(), tgt.data_ptr(), idt2.data_ptr(), g_sums.data_ptr(), B, S, H, W,
                                              out3.data_ptr(), aux3.data_ptr(), sums3.data_ptr(), gw3.data_ptr() if want_gw else 0,
                                              gd3.data_ptr(), gp3.data_ptr(), ws3.data_ptr(), n3, st),
                "ds_fused_loss_fb_warp_bwd")
        torch.cuda.synchronize()
        assert torch.equal(out3, out1) and torch.equal(aux3 & 0x1f, aux1 & 0x1f)
        assert float(((sums3 - sums1).abs() / sums1.abs()).max()) < 1e-6
        assert torch.equal(gd3, gd1), float((gd3 - gd1).abs().max())
        assert rel_to_max(N(gp3), N(gp1)) < 1e-4
        if want_gw:
            assert torch.equal(gw3, gw1)
        else:
            assert bool(torch.isnan(gw3).all())        # left untouched
    rc = lib.ds_fused_loss_fb(warped.data_ptr(), tgt.data_ptr(), idt2.data_ptr(), 0, B, 3, H, W, out2.data_ptr(), aux2.data_ptr(),
                              sums2.data_ptr(), gw2.data_ptr(), ws2.data_ptr(), n2, st)
    assert rc == -2 or (rc == -1 and S != 3)     # three sources: unsupported (bad_dims first only if the planes overflow)


@pytest.mark.parametrize("B,S,H,W", [(12, 2, 192, 640), (3, 1, 37, 131), (2, 2, 5, 3), (1, 2, 2, 2), (2, 2, 23, 64), (2, 2, 64, 65),
                                     (2, 3, 40, 72), (1, 4, 33, 130), (1, 2, 3, 200), (5, 2, 96, 320), (1, 2, 1080, 1920)])
def test_warp_and_identity_in_one_launch(de, B, S, H, W):
    """ds_warp_identity_fwd: the warp and the identity loss map of train.py:85 from ONE launch -- for S <= 2 the waves of
    the loss workgroups warp the band's rows themselves (E wave: source 0, channel waves: the last source, gathers issued
    a step ahead), for S > 2 warp workgroups run beside loss workgroups -- against the two entry points it replaces,
    bit for bit: warped == ds_warp_fwd, idt_loss == the identity map of ds_fused_loss_fwd.  Sizes below, at and off the
    64-column band, the 2-row unit and the 4-row slab, one- to four-row images."""
    L = de._lib
    lib, st = L.lib(), L.stream()
    g = torch.Generator().manual_seed(B * 100 + S * 10 + H + W)
    N_ = B * S
    src = torch.rand(B, S, 3, H, W, generator=g).to(DEV)
    tgt = torch.rand(B, 3, H, W, generator=g).to(DEV)
    depth = (1 + 79 * torch.rand(B, S, H, W, generator=g)).to(DEV)
    vec = (0.02 * torch.randn(N_, 6, generator=g)).to(DEV)
    pose = de.synthesis.construct_pose_batch(vec, DEV).reshape(B, S, 3, 4).contiguous()
    K, Kinv = [t.to(DEV) for t in de.fused.intrinsics(W, H)]
    warped1 = torch.full((B, S, 3, H, W), float("nan"), device=DEV)
    idt1 = torch.full((B, H, W), float("nan"), device=DEV)
    L.check(lib.ds_warp_identity_fwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                     tgt.data_ptr(), B, S, H, W, warped1.data_ptr(), idt1.data_ptr(), st), "ds_warp_identity_fwd")
    warped0 = torch.empty_like(warped1)
    L.check(lib.ds_warp_fwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(), N_, 3, H, W, H, W,
                            warped0.data_ptr(), st), "ds_warp_fwd")
    out0 = torch.empty(B, H, W, device=DEV)
    idt0 = torch.empty(B, H, W, device=DEV)
    aux0 = torch.empty(B, H, W, dtype=torch.uint8, device=DEV)
    sums0 = torch.empty(B, 2, device=DEV)
    n = lib.ds_fused_loss_workspace_bytes(B, H, W)
    ws = torch.empty(max(n, 4), dtype=torch.uint8, device=DEV)
    L.check(lib.ds_fused_loss_fwd(warped0.data_ptr(), src.data_ptr(), tgt.data_ptr(), B, S, H, W, out0.data_ptr(), idt0.data_ptr(),
                                  aux0.data_ptr(), sums0.data_ptr(), ws.data_ptr(), n, st), "ds_fused_loss_fwd")
    torch.cuda.synchronize()
    assert torch.equal(warped1, warped0)
    assert torch.equal(idt1, idt0)


@pytest.mark.parametrize("B,S,H,W", [(12, 2, 192, 640), (3, 3, 37, 131), (2, 1, 9, 70), (1, 2, 2, 2), (2, 4, 26, 64), (1, 2, 375, 1242)])
def test_train_step_in_one_call(de, oracle, B, S, H, W):
    """ds_train_step_fwdbwd (train.py:72-89 and its autograd in one C call: ds_loss_step_fwdbwd + the smoothness value and
    gradient from one kernel + total) against the calls it replaces: maps, sums, g_pose bit for bit (the same launches),
    g_depth bit for bit (the smoothness gradient is the same arithmetic added to the same photometric gradient), reg to
    1e-6 relative (another order of the partial sums) and against the oracle's regularization_term (1e-5), total ==
    sum_b sums[b,0] + scale * reg to 1e-6."""
    L = de._lib
    lib, st = L.lib(), L.stream()
    g = torch.Generator().manual_seed(B * 7 + S * 3 + H + W)
    N_ = B * S
    src = torch.rand(B, S, 3, H, W, generator=g).to(DEV)
    tgt = torch.rand(B, 3, H, W, generator=g).to(DEV)
    depth = (1 + 79 * torch.rand(B, S, H, W, generator=g)).to(DEV)
    vec = (0.02 * torch.randn(N_, 6, generator=g)).to(DEV)
    pose = de.synthesis.construct_pose_batch(vec, DEV).reshape(B, S, 3, 4).contiguous()
    K, Kinv = [t.to(DEV) for t in de.fused.intrinsics(W, H)]
    scale = 0.02
    f = dict(dtype=torch.float32, device=DEV)

    def bufs():
        return dict(warped=torch.empty(B, S, 3, H, W, **f), out=torch.empty(B, H, W, **f), idt=torch.empty(B, H, W, **f),
                    aux=torch.empty(B, H, W, dtype=torch.uint8, device=DEV), sums=torch.empty(B, 2, **f),
                    gw=torch.empty(B, S, 3, H, W, **f), gd=torch.empty(B, S, H, W, **f), gp=torch.empty(N_, 3, 4, **f))
    a, b_ = bufs(), bufs()
    reg, total = torch.empty(1, **f), torch.empty(1, **f)
    n = lib.ds_train_step_fwdbwd_workspace_bytes(B, S, H, W)
    ws = torch.empty(n, dtype=torch.uint8, device=DEV)
    L.check(lib.ds_train_step_fwdbwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                     tgt.data_ptr(), scale, B, S, H, W, a["warped"].data_ptr(), a["out"].data_ptr(),
                                     a["idt"].data_ptr(), a["aux"].data_ptr(), a["sums"].data_ptr(), reg.data_ptr(),
                                     total.data_ptr(), a["gw"].data_ptr(), a["gd"].data_ptr(), a["gp"].data_ptr(), ws.data_ptr(),
                                     n, st), "ds_train_step_fwdbwd")
    # the calls it replaces
    n2 = max(lib.ds_loss_step_fwdbwd_workspace_bytes(B, S, H, W), lib.ds_smooth_workspace_bytes(N_, H, W))
    ws2 = torch.empty(n2, dtype=torch.uint8, device=DEV)
    L.check(lib.ds_loss_step_fwdbwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                    tgt.data_ptr(), 0, B, S, H, W, b_["warped"].data_ptr(), b_["out"].data_ptr(),
                                    b_["idt"].data_ptr(), b_["aux"].data_ptr(), b_["sums"].data_ptr(), b_["gw"].data_ptr(),
                                    b_["gd"].data_ptr(), b_["gp"].data_ptr(), 0, ws2.data_ptr(), n2, st), "ds_loss_step_fwdbwd")
    reg2 = torch.empty(1, **f)
    L.check(lib.ds_smooth_batch_fwd(depth.data_ptr(), tgt.data_ptr(), B, S, 3, H, W, reg2.data_ptr(), ws2.data_ptr(), n2, st),
            "ds_smooth_batch_fwd")
    g_reg = torch.full((1,), scale, **f)
    L.check(lib.ds_smooth_batch_bwd(depth.data_ptr(), tgt.data_ptr(), g_reg.data_ptr(), B, S, 3, H, W, 1, b_["gd"].data_ptr(), st),
            "ds_smooth_batch_bwd")
    torch.cuda.synchronize()
    for k in ("warped", "out", "idt", "aux", "sums", "gp", "gd"):
        assert torch.equal(a[k], b_[k]), k
    assert abs(float(reg) - float(reg2)) <= 1e-6 * abs(float(reg2))
    want = float(b_["sums"][:, 0].sum()) + scale * float(reg2)
    assert abs(float(total) - want) <= 1e-6 * abs(want)
    # without the d loss / d warped scratch: fine where it never crosses memory (S <= 3), an argument error otherwise
    c = bufs()
    rc = lib.ds_train_step_fwdbwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                  tgt.data_ptr(), scale, B, S, H, W, c["warped"].data_ptr(), c["out"].data_ptr(),
                                  c["idt"].data_ptr(), c["aux"].data_ptr(), c["sums"].data_ptr(), reg.data_ptr(),
                                  total.data_ptr(), 0, c["gd"].data_ptr(), c["gp"].data_ptr(), ws.data_ptr(), n, st)
    torch.cuda.synchronize()
    if S <= 3:
        assert rc == 0
        for k in ("warped", "out", "idt", "aux", "sums", "gp", "gd"):
            assert torch.equal(a[k], c[k]), k
    else:
        assert rc == -1
    # regularization_term of the reference (oracle): depth maps (N,1,H,W) against their targets stacked S times
    tg = tgt[:, None].expand(B, S, 3, H, W).reshape(N_, 3, H, W)
    ref = oracle.regularization_term(N(depth.reshape(N_, 1, H, W)), N(tg))
    assert abs(float(reg) - float(ref)) <= 1e-5 * abs(float(ref))


def test_fused_step_differential_fuzz(de):
    """scripts/fuzz_step.py: 80 random (B, S, H, W) with ties between sources, static scenes, near and far depths, small and
    large poses -- ds_train_step_fwdbwd against the unfused kernels: maps bit for bit, gradients to 1e-5 of the largest."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import fuzz_step
    assert fuzz_step.run(80, seed=11) == 0


def test_reprojection_loss_argument_forms(de, oracle):
    """metrics.reprojection_loss takes what the reference takes (metrics.py:29-30,62-72): a materialised target stack
    (train.py:80; one launch per source, no value is read back to the host), an expanded view or a single target image (one
    launch), rank-3 inputs, one prediction against S targets -- against reference-made vectors (tests/golden/
    facade_vectors.npz, tools/gen_golden.py facade) bit for bit, and a stack whose slices differ against the oracle."""
    gold = load_golden("facade_vectors")
    d = load_golden("synth_oov_24x80")
    warped, target, tstack = T(d["warped"]), T(d["target"]), T(gold["tstack"])
    S, _, H, W = warped.shape
    stack = torch.stack([target] * S)
    forms = [de.metrics.reprojection_loss(warped, stack),                              # train.py:80,84
             de.metrics.reprojection_loss(warped, stack, shared_target=True),
             de.metrics.reprojection_loss(warped, target[None].expand(S, -1, -1, -1)),
             de.metrics.reprojection_loss(warped, target),                             # (S,C,H,W) against (C,H,W)
             de.metrics.reprojection_loss(warped, target[None])]
    for f in forms:
        assert f.shape == (1, 1, H, W) and np.array_equal(N(f), d["out_loss"]) and np.array_equal(N(f), gold["rank3_target"])
    assert np.array_equal(N(de.metrics.reprojection_loss(warped[0], target)), gold["rank3_both"])
    assert np.array_equal(N(de.metrics.reprojection_loss(warped[0], tstack)), gold["rank3_pred"])
    got = de.metrics.reprojection_loss(warped, tstack)
    assert np.array_equal(N(got), gold["rank4_distinct_targets"])
    want = np.minimum.reduce([oracle.reprojection_loss(N(warped[s:s + 1]), N(tstack[s]))[0] for s in range(S)])
    assert np.array_equal(N(got).reshape(H, W), want)
    assert np.array_equal(N(de.metrics.reprojection_error(warped[0], target)), gold["err_rank3"])
    # in-place changes of a stack are seen (nothing is remembered between calls)
    stack[1].mul_(0.5)
    c2 = de.metrics.reprojection_loss(warped, stack)
    want = np.minimum.reduce([oracle.reprojection_loss(N(warped[s:s + 1]), N(stack[s]))[0] for s in range(S)])
    assert np.array_equal(N(c2).reshape(H, W), want)
    # gradients flow through every form (rank-3 prediction against S targets: summed over the broadcast)
    p = warped[0].clone().requires_grad_(True)
    de.metrics.reprojection_loss(p, tstack).sum().backward()
    q = warped[0][None].expand(S, -1, -1, -1).clone().requires_grad_(True)
    de.metrics.reprojection_loss(q, tstack).sum().backward()
    assert rel_to_max(N(p.grad), N(q.grad.sum(0))) < 1e-6


def test_rmsle_eps(de, oracle):
    """metrics.rmsle(pred, gt, eps) for eps other than the default (metrics.py:114,122) against the reference's values
    (facade_vectors.npz) and the oracle, 1e-4 relative."""
    gold = load_golden("facade_vectors")
    d = load_golden("metrics_3x40x96")
    pred, gt = T(d["pred"]), T(d["gt"])
    for tag, eps in (("1e-3", 1e-3), ("0.5", 0.5), ("0", 0.0)):
        v = float(de.metrics.rmsle(pred, gt, eps=eps))
        want = float(gold["rmsle_eps_" + tag])
        assert abs(v - want) < 1e-4 * abs(want), tag
        assert abs(v - oracle.depth_metrics(d["pred"], d["gt"], eps=eps)["rmsle"]) < 1e-5 * abs(want)
    assert abs(float(de.metrics.rmsle(pred, gt)) - float(d["rmsle"])) < 1e-4 * float(d["rmsle"])


def test_one_call_step_with_image_gradient(de):
    """ds_loss_step_fwdbwd with g_src: the gradient w.r.t. the source images through the warp (cleared inside the call)
    equals autograd through fused_reprojection_loss(image_grad=True) -- the same kernels on the same d loss / d warped, and
    no float atomics: bit for bit -- as do depth and pose gradients; called twice in a row on a buffer holding stale
    values to show that every texel is written by the call."""
    L = de._lib
    lib, st = L.lib(), L.stream()
    for B, S, H, W in ((3, 2, 40, 72), (2, 3, 37, 131)):       # S = 3 takes the unfused kernels
        g = torch.Generator().manual_seed(B * 31 + W)
        src = torch.rand(B, S, 3, H, W, generator=g).to(DEV)
        tgt = torch.rand(B, 3, H, W, generator=g).to(DEV)
        depth = (1 + 79 * torch.rand(B, S, 1, H, W, generator=g)).to(DEV)
        pose = de.ops.pose_from_vec((0.01 * torch.randn(B * S, 6, generator=g)).to(DEV)).reshape(B, S, 3, 4)
        K, Kinv = [t.to(DEV) for t in de.fused.intrinsics(W, H)]
        s1, d1, p1 = src.clone().requires_grad_(True), depth.clone().requires_grad_(True), pose.clone().requires_grad_(True)
        r = de.fused.fused_reprojection_loss(s1, tgt, d1, p1, K, Kinv, image_grad=True)
        r["loss"].sum().backward()
        f32 = dict(dtype=torch.float32, device=DEV)
        warped, gw = torch.empty(B, S, 3, H, W, **f32), torch.empty(B, S, 3, H, W, **f32)
        out, idt = torch.empty(B, H, W, **f32), torch.empty(B, H, W, **f32)
        aux, sums = torch.empty(B, H, W, dtype=torch.uint8, device=DEV), torch.empty(B, 2, **f32)
        gd, gp = torch.empty(B, S, H, W, **f32), torch.empty(B, S, 3, 4, **f32)
        gs = torch.full((B, S, 3, H, W), 7.0, **f32)           # stale contents: the call must clear them
        n = lib.ds_loss_step_fwdbwd_image_workspace_bytes(B, S, H, W)
        ws = torch.empty(n, dtype=torch.uint8, device=DEV)
        for _ in range(2):
            L.check(lib.ds_loss_step_fwdbwd(K.data_ptr(), Kinv.data_ptr(), pose.data_ptr(), depth.data_ptr(), src.data_ptr(),
                                            tgt.data_ptr(), 0, B, S, H, W, warped.data_ptr(), out.data_ptr(), idt.data_ptr(),
                                            aux.data_ptr(), sums.data_ptr(), gw.data_ptr(), gd.data_ptr(), gp.data_ptr(),
                                            gs.data_ptr(), ws.data_ptr(), n, st), "ds_loss_step_fwdbwd")
        torch.cuda.synchronize()
        assert torch.equal(out, r["out_loss"]) and torch.equal(warped, r["warped"])
        assert torch.equal(gs, s1.grad)
        assert rel_to_max(N(gd), N(d1.grad)[:, :, 0]) < 1e-6
        assert rel_to_max(N(gp), N(p1.grad)) < 1e-4


def test_eager_step_matches_the_two_call_step(de):
    """ds_loss_step_fwdbwd (forward + backward launches and one merged finalize in one call; gradients formed for a unit
    upstream gradient, scaled in backward) == the forward call / backward call pair: maps and means bit for bit (same
    kernels, same partial-sum layout), gradients under a non-unit upstream gradient to 1e-4 of the largest (BASELINE's
    gradient bar; the factor enters before the warp backward's cancelling sums in one form and after them in the other:
    1e-5 observed); B=12 192x640 and ragged."""
    for B, S, H, W in ((12, 2, 192, 640), (3, 3, 37, 131), (2, 4, 26, 64), (1, 1, 2, 2), (1, 2, 1080, 1920), (2, 1, 375, 1242)):
        g = torch.Generator().manual_seed(B * 100 + H)
        src = torch.rand(B, S, 3, H, W, generator=g).to(DEV)
        tgt = torch.rand(B, 3, H, W, generator=g).to(DEV)
        depth = (1 + 79 * torch.rand(B, S, 1, H, W, generator=g)).to(DEV)
        vec = (0.01 * torch.randn(B * S, 6, generator=g)).to(DEV)
        K, Kinv = [t.to(DEV) for t in de.fused.intrinsics(W, H)]
        res = []
        for eager in (False, True):
            d, v = depth.clone().requires_grad_(True), vec.clone().requires_grad_(True)
            r = de.fused.train_step_loss(src, tgt, d, de.ops.pose_from_vec(v).reshape(B, S, 3, 4), K, Kinv, eager_grads=eager)
            (r["total"] * 1.7).backward()
            res.append((r, d.grad, v.grad))
        (a, gda, gva), (b, gdb, gvb) = res
        for k in ("warped", "out_loss", "idt_loss", "aux", "reg"):
            assert torch.equal(a[k], b[k]), (k, B, H, W)
        for k in ("loss", "log"):   # per-workgroup fp32 partials, grouped by each kernel's own bands and row chunks
            assert float(((a[k] - b[k]).abs() / b[k].abs()).max()) < 1e-6, (k, B, H, W)
        assert abs(float(a["total"]) - float(b["total"])) < 1e-6 * abs(float(b["total"]))
        assert rel_to_max(N(gdb), N(gda)) < 1e-4 and rel_to_max(N(gvb), N(gva)) < 1e-4


def test_ingest_u8_matches_to_tensor(de):
    """to_tensor of the resized frames on the device == the float tensors the reference computed from the
    same uint8 frames (dataset.py:51; fixtures real_*_frames_u8.npz and the src/target of real_*.npz), bit-exact;
    plus a ragged size (H*W not a multiple of 4) against value/255 and the (H,W,3) single-frame form."""
    for name in ("real_192x640", "real_48x160"):
        fr = np.load(os.path.join(os.path.dirname(__file__), "golden", name + "_frames_u8.npz"))["frames"]
        d = load_golden(name)
        got = N(de.ingest.to_tensor(torch.from_numpy(fr).to(DEV)))
        assert got.shape == (fr.shape[0], 3) + fr.shape[1:3]
        assert np.array_equal(got[:3], d["src"]) and np.array_equal(got[3], d["target"])
    g = torch.Generator().manual_seed(5)
    odd = torch.randint(0, 256, (2, 7, 9, 3), generator=g, dtype=torch.uint8)
    want = (odd.float() / 255).permute(0, 3, 1, 2).numpy()
    assert np.array_equal(N(de.ingest.to_tensor(odd.to(DEV))), want)
    assert np.array_equal(N(de.ingest.to_tensor(odd[1].to(DEV))), want[1])
    with pytest.raises(RuntimeError):
        de.ingest.to_tensor(odd)                      # CPU tensor
    with pytest.raises(RuntimeError):
        de.ingest.to_tensor(odd.to(DEV).float())      # wrong dtype


def test_ingest_resize_matches_pil(de, oracle):
    """SURVEY 8f.4 with the resize on the device: tf.to_tensor(tf.resize(img, (192, 640))) (dataset.py:51,110) from the RAW
    uint8 frame == the frames / float tensors the reference-generated fixtures hold, bit-exact (Pillow's fixed-point
    arithmetic is integer work); ragged sizes, up- and down-scaling, batches, against the oracle (itself pinned to PIL)."""
    raw = np.load(os.path.join(os.path.dirname(__file__), "golden", "raw_frame_376x1241_u8.npz"))["frame"]
    for name, size in (("real_192x640", (192, 640)), ("real_48x160", (48, 160))):
        want_u8 = np.load(os.path.join(os.path.dirname(__file__), "golden", name + "_frames_u8.npz"))["frames"][0]
        d = load_golden(name)
        got_u8 = de.ingest.resize_u8(torch.from_numpy(raw).to(DEV), size)
        assert got_u8.dtype == torch.uint8 and np.array_equal(N(got_u8), want_u8)
        got_f = de.ingest.resize_to_tensor(torch.from_numpy(raw).to(DEV), size)
        assert np.array_equal(N(got_f), d["src"][0])              # frame 0 is the first source of the fixture
        assert np.array_equal(N(de.ingest.to_tensor(got_u8)), d["src"][0])
    rng = np.random.default_rng(1)
    for H, W, ho, wo in ((37, 53, 19, 20), (19, 20, 37, 53), (64, 64, 64, 64), (90, 7, 3, 50), (5, 5, 11, 2), (1, 1, 4, 4),
                         (400, 300, 13, 17), (33, 1000, 33, 40), (376, 1241, 96, 320)):
        a = rng.integers(0, 256, (3, H, W, 3), dtype=np.uint8)
        want = np.stack([oracle.pil_resize_bilinear_u8(f, (ho, wo)) for f in a])
        got = de.ingest.resize_u8(torch.from_numpy(a).to(DEV), (ho, wo))
        assert np.array_equal(N(got), want), (H, W, ho, wo)
        gf = de.ingest.resize_to_tensor(torch.from_numpy(a).to(DEV), (ho, wo))
        assert np.array_equal(N(gf), (want.astype(np.float32) / np.float32(255)).transpose(0, 3, 1, 2))
    with pytest.raises(RuntimeError):
        de.ingest.resize_u8(torch.zeros(100, 7, 3, dtype=torch.uint8, device=DEV), (3, 50))   # 33x down: unsupported
    with pytest.raises(RuntimeError):
        de.ingest.resize_u8(torch.zeros(8, 8, 3, dtype=torch.uint8), (4, 4))                  # CPU tensor


def test_ingest_depth_u16(de):
    """dataset.py:116-120: 16-bit ground-truth samples / 2**16.  The reference holds no depth PNG and torchvision is not
    installed: parity unpinned, checked bit-exact against the direct statement on every 16-bit value and a ragged size."""
    allv = np.arange(65536, dtype=np.uint16).reshape(1, 256, 256)
    want = allv.astype(np.float32) / np.float32(65536)
    got = de.ingest.depth_to_tensor(torch.from_numpy(allv).to(DEV))
    assert got.shape == (1, 1, 256, 256) and np.array_equal(N(got)[:, 0], want)
    rng = np.random.default_rng(3)
    odd = rng.integers(0, 65536, (3, 7, 9), dtype=np.uint16)            # odd element count: the tail sample
    want = odd.astype(np.float32) / np.float32(65536)
    assert np.array_equal(N(de.ingest.depth_to_tensor(torch.from_numpy(odd).to(DEV)))[:, 0], want)
    assert np.array_equal(N(de.ingest.depth_to_tensor(torch.from_numpy(odd[2]).to(DEV)))[0], want[2])
    assert np.array_equal(N(de.ingest.depth_to_tensor(torch.from_numpy(odd.astype(np.int32)).to(DEV)))[:, 0], want)   # PIL mode I
    assert np.array_equal(N(de.ingest.depth_to_tensor(torch.from_numpy(odd.view(np.int16)).to(DEV)))[:, 0], want)
    with pytest.raises(RuntimeError):
        de.ingest.depth_to_tensor(torch.from_numpy(odd))
    with pytest.raises(RuntimeError):
        de.ingest.depth_to_tensor(torch.full((2, 2), 70000, dtype=torch.int32, device=DEV))


def test_rejects_cpu_tensors_and_bad_shapes(de):
    with pytest.raises(RuntimeError):
        de.metrics.rmse(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4))
    with pytest.raises(RuntimeError):
        de.metrics.reprojection_loss(torch.zeros(2, 3, 8, 8, device=DEV), torch.zeros(2, 3, 8, 9, device=DEV))
    with pytest.raises(RuntimeError):
        de.metrics.reprojection_error(torch.zeros(2, 3, 8, 8, device=DEV, dtype=torch.float64),
                                      torch.zeros(2, 3, 8, 8, device=DEV, dtype=torch.float64))


@pytest.mark.parametrize("B,S,H,W", [(12, 2, 192, 640), (2, 4, 50, 131), (3, 2, 37, 131), (2, 1, 9, 70), (1, 2, 5, 3),
                                     (2, 3, 26, 64), (1, 1, 2, 2), (1, 2, 3, 129), (2, 2, 17, 63)])
def test_full_size_against_oracle_and_properties(de, oracle, B, S, H, W):
    """BASELINE config 2/3 size (and ragged cases: S=4 and S=3 take the DPP forward kernel, S<=2 the channel-split one;
    widths below, at and off multiples of the 64-column band): every sample against the oracle, plus
    size-independent properties: tie-breaking, linearity of the backward, batch independence."""
    g = torch.Generator().manual_seed(123)
    src = torch.rand(B, S, 3, H, W, generator=g)
    tgt = torch.rand(B, 3, H, W, generator=g)
    depth = 1 + 79 * torch.rand(B, S, 1, H, W, generator=g)
    vec = 0.01 * torch.randn(B * S, 6, generator=g)
    K, Kinv = de.fused.intrinsics(W, H)
    pose_np = oracle.pose_from_vec(vec.numpy()).reshape(B, S, 3, 4)
    tie = B > 1 and S > 1
    if tie:
        src[1, 1] = src[1, 0]; depth[1, 1] = depth[1, 0]; pose_np[1, 1] = pose_np[1, 0]   # a tie: first index must win
    dd = depth.to(DEV).requires_grad_(True)
    pp = T(pose_np).requires_grad_(True)
    r = de.fused.fused_reprojection_loss(src.to(DEV), tgt.to(DEV), dd, pp, K.to(DEV), Kinv.to(DEV))
    (2.0 * r["loss"].sum()).backward()
    if tie:
        assert not np.any((N(r["aux"])[1] & 0x1f) == 1)   # sources 0 and 1 of sample 1 are identical
    Kn, Kin = K.numpy(), Kinv.numpy()
    for b in range(B):
        ow = np.stack([oracle.warp(Kn, Kin, pose_np[b, s], depth[b, s].numpy(), src[b, s].numpy(), H, W) for s in range(S)])
        assert np.array_equal(N(r["warped"])[b], ow)
        la = oracle.loss_assembly(ow, src[b].numpy(), tgt[b].numpy())
        assert np.array_equal(N(r["out_loss"])[b], la["out_loss"])
        assert np.array_equal(N(r["idt_loss"])[b], la["idt_loss"])
        assert np.array_equal(N(r["aux"])[b] & 0x1f, la["argmin"])
        assert abs(float(r["loss"][b]) - la["mean_ratio"]) < 1e-5
        if b < 3:
            gw = oracle.loss_assembly_bwd(ow, tgt[b].numpy(), la["idt_loss"], la["argmin"], g_scale=2.0)
            gd, gp, _ = zip(*[oracle.warp_bwd(Kn, Kin, pose_np[b, s], depth[b, s].numpy(), src[b, s].numpy(), gw[s], H, W, False)
                              for s in range(S)])
            # 1e-4 of the largest gradient (BASELINE tolerance); a handful of pixels (the 2x2 case) has nothing to
            # average the fp32 rounding of two differently ordered evaluations over: 5e-4 there
            tol = 1e-4 if H * W >= 64 else 5e-4
            assert rel_to_max(N(dd.grad)[b, :, 0], np.stack(gd)) < tol
            assert rel_to_max(N(pp.grad)[b], np.stack(gp)) < tol


def test_config4_size_partition_invariance_and_spot_oracle(de, oracle):
    """BASELINE config 4 (B=64, S=2, 320x1024, 1.8 GB working set): the launch geometry depends on the batch size, the
    results must not -- the whole batch == the batch run as 40 + 24 targets, bit for bit (maps, depth and pose gradients; the
    per-target means to 1e-6); first and last target against the oracle; depth metrics of the batch against the oracle."""
    g = torch.Generator().manual_seed(404)
    B, S, H, W = 64, 2, 320, 1024
    src = torch.rand(B, S, 3, H, W, generator=g)
    tgt = torch.rand(B, 3, H, W, generator=g)
    depth = 1 + 79 * torch.rand(B, S, 1, H, W, generator=g)
    vec = 0.01 * torch.randn(B * S, 6, generator=g)
    K, Kinv = de.fused.intrinsics(W, H)
    pose_np = oracle.pose_from_vec(vec.numpy()).reshape(B, S, 3, 4)
    Kd, Kid, srcd, tgtd = K.to(DEV), Kinv.to(DEV), src.to(DEV), tgt.to(DEV)

    def run(lo, hi):
        dd = depth[lo:hi].to(DEV).requires_grad_(True)
        pp = T(pose_np[lo:hi]).requires_grad_(True)
        r = de.fused.fused_reprojection_loss(srcd[lo:hi], tgtd[lo:hi], dd, pp, Kd, Kid)
        r["loss"].sum().backward()
        return r, dd.grad, pp.grad

    full, gd, gp = run(0, B)
    for lo, hi in ((0, 40), (40, B)):
        part, pgd, pgp = run(lo, hi)
        for k in ("warped", "out_loss", "idt_loss", "aux"):
            assert torch.equal(full[k][lo:hi], part[k]), (k, lo)
        for k in ("loss", "log"):  # per-wave fp32 partials: the row chunks, and with them the grouping, follow the batch size
            assert float(((full[k][lo:hi] - part[k]).abs() / part[k].abs()).max()) < 1e-6, (k, lo)
        assert torch.equal(gd[lo:hi], pgd) and torch.equal(gp[lo:hi], pgp)
        del part, pgd, pgp
    Kn, Kin = K.numpy(), Kinv.numpy()
    for b in (0, B - 1):
        ow = np.stack([oracle.warp(Kn, Kin, pose_np[b, s], depth[b, s].numpy(), src[b, s].numpy(), H, W) for s in range(S)])
        assert np.array_equal(N(full["warped"][b]), ow)
        la = oracle.loss_assembly(ow, src[b].numpy(), tgt[b].numpy())
        assert np.array_equal(N(full["out_loss"][b]), la["out_loss"]) and np.array_equal(N(full["aux"][b]) & 0x1f, la["argmin"])
        assert abs(float(full["loss"][b]) - la["mean_ratio"]) < 1e-5
    gt = depth[:, 0].clone()
    gt[torch.rand(gt.shape, generator=g) < 0.3] = 0.0
    pred = depth[:, 1].contiguous()
    m = de.metrics.depth_metrics(pred.to(DEV), gt.to(DEV))
    o = oracle.depth_metrics(pred.numpy(), gt.numpy())
    for k in ("rmse", "rmsle", "abs_rel", "sq_rel", "delta1", "n_valid"):
        assert abs(float(m[k]) - o[k]) < 1e-4 * max(abs(o[k]), 1e-12), k
