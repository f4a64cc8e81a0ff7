"""Pins the CPU oracle (oracle/ds_oracle.c) against the golden vectors captured from the reference.

CPU-only.  Bit-exact (CRC32 of the raw fp32 bytes) for every forward stage that has no
transcendental in it; 1e-4 relative-to-max for gradients and scalars (BASELINE.json tolerance).
"""
import os

import numpy as np
import pytest

from conftest import LOSS_CASES, crc32, load_golden, real_frames, rel_to_max


@pytest.mark.parametrize("name", ["pose_batch"])
def test_pose_from_vec(oracle, name):
    d = load_golden(name)
    pm = oracle.pose_from_vec(d["vec"])
    # sinf/cosf/norm of libm vs torch: last-bit differences on the large-angle row only
    assert np.abs(pm - d["pose_mats"]).max() <= 2e-7
    assert np.array_equal(pm[4], np.concatenate([np.eye(3), np.zeros((3, 1))], 1).astype(np.float32))  # zero vector


@pytest.mark.parametrize("name", LOSS_CASES)
def test_forward_stages_bit_exact(oracle, name):
    d = load_golden(name)
    src, tgt, depth, K, Kinv = d["src"], d["target"], d["depth"], d["K"], d["Kinv"]
    S, C, H, W = src.shape
    pm = oracle.pose_from_vec(d["pose_vec"])
    assert crc32(pm) == int(d["crc_pose_mats"])
    coords = np.stack([oracle.reproject(K, Kinv, pm[s], depth[s], H, W) for s in range(S)])
    assert crc32(coords) == int(d["crc_coords_px"])                      # synthesis.py:83-116
    warped, norm = zip(*[oracle.grid_sample(src[s], coords[s]) for s in range(S)])
    warped, norm = np.stack(warped), np.stack(norm)
    assert crc32(norm) == int(d["crc_coords_norm"])                      # in-place normalisation, synthesis.py:50-51
    assert crc32(warped) == int(d["crc_warped"])                         # grid_sample, synthesis.py:54
    fused = np.stack([oracle.warp(K, Kinv, pm[s], depth[s], src[s], H, W) for s in range(S)])
    assert np.array_equal(fused, warped)
    ssim = oracle.ssim(warped, np.stack([tgt] * S))
    assert crc32(ssim) == int(d["crc_ssim_warped"])                      # metrics.py:21-51
    la = oracle.loss_assembly(warped, src, tgt)
    assert crc32(la["out_loss"]) == int(d["crc_out_loss"])               # metrics.py:54-72
    assert crc32(la["idt_loss"]) == int(d["crc_idt_loss"])
    assert np.array_equal(la["argmin"], d["argmin"].reshape(H, W))
    out, am, per = oracle.reprojection_loss(warped, tgt)
    assert crc32(per) == int(d["crc_per_src_loss"])
    assert np.abs(la["ratio"] - d["ratio"].reshape(H, W)).max() <= 1e-6  # expf: not bit-exact, train.py:88
    assert abs(la["mean_ratio"] - float(d["loss_photo"])) <= 1e-6
    assert abs(la["mean_out"] - float(d["log_val"])) <= 1e-6


@pytest.mark.parametrize("name", LOSS_CASES)
def test_backward_matches_reference_autograd(oracle, name):
    d = load_golden(name)
    src, tgt, depth, K, Kinv, pm = d["src"], d["target"], d["depth"], d["K"], d["Kinv"], d["pose_mats"]
    S, C, H, W = src.shape
    gw = oracle.loss_assembly_bwd(d["warped"], tgt, d["idt_loss"].reshape(H, W), d["argmin"].reshape(H, W))
    assert rel_to_max(gw, d["g_warped"]) < 1e-4
    val, greg = oracle.regularization_term(depth / np.float32(80.0), np.stack([tgt] * S), want_grad=True)
    assert abs(val - float(d["reg"])) < 1e-4 * abs(float(d["reg"]))       # metrics.py:75-97
    assert rel_to_max(greg / 80.0, d["g_depth_reg"]) < 1e-4
    gd, gp, gs = zip(*[oracle.warp_bwd(K, Kinv, pm[s], depth[s], src[s], gw[s], H, W) for s in range(S)])
    g_depth = np.stack(gd)[:, None] + 0.02 * greg / 80.0                  # train.py:89
    assert rel_to_max(g_depth, d["g_depth"]) < 1e-4
    assert rel_to_max(np.stack(gp), d["g_pose_mats"]) < 1e-4
    assert rel_to_max(np.stack(gs), d["g_src"]) < 1e-4


def test_grid_sample_bwd_consistent_with_warp_bwd(oracle):
    d = load_golden("synth_oov_24x80")
    s = 0
    H, W = d["src"].shape[-2:]
    gg, gs = oracle.grid_sample_bwd(d["src"][s], d["coords_px"][s], d["g_warped"][s])
    _, _, gs2 = oracle.warp_bwd(d["K"], d["Kinv"], d["pose_mats"][s], d["depth"][s], d["src"][s], d["g_warped"][s], H, W)
    assert rel_to_max(gs, gs2) < 1e-5
    assert np.isfinite(gg).all()
    # out-of-view pixels (un-clipped coordinate on/outside the border) get zero grid gradient
    u = d["coords_px"][s][..., 0]
    assert np.all(gg[..., 0][(u <= 0) | (u >= W - 1)] == 0)


def test_config1_full_resolution_192x640(oracle):
    """BASELINE config 1: test_data_copy frames 0-3 at 192x640, S=3 (strided samples + CRCs stored)."""
    d = load_golden("real_192x640")
    fr = real_frames("real_192x640")
    src, tgt = fr[:-1], fr[-1]
    assert np.array_equal(src, d["src"]) and np.array_equal(tgt, d["target"])
    depth, K, Kinv = d["depth"], d["K"], d["Kinv"]
    S, C, H, W = src.shape
    sub = int(d["sub"])
    pm = oracle.pose_from_vec(d["pose_vec"])
    assert crc32(pm) == int(d["crc_pose_mats"])
    warped = np.stack([oracle.warp(K, Kinv, pm[s], depth[s], src[s], H, W) for s in range(S)])
    assert crc32(warped) == int(d["crc_warped"])
    assert np.array_equal(warped[..., ::sub, ::sub], d["sub_warped"])
    la = oracle.loss_assembly(warped, src, tgt)
    assert crc32(la["out_loss"]) == int(d["crc_out_loss"])
    assert crc32(la["idt_loss"]) == int(d["crc_idt_loss"])
    assert crc32(la["argmin"]) == int(d["crc_argmin"])
    assert abs(la["mean_ratio"] - float(d["loss_photo"])) <= 1e-6
    gw = oracle.loss_assembly_bwd(warped, tgt, la["idt_loss"], la["argmin"])
    assert rel_to_max(gw[..., ::sub, ::sub], d["sub_g_warped"]) < 1e-4
    assert abs(gw.astype(np.float64).sum() - float(d["sum_g_warped"])) < 1e-4 * float(d["abssum_g_warped"])
    val, greg = oracle.regularization_term(depth / np.float32(80.0), np.stack([tgt] * S), want_grad=True)
    gd, gp, gs = zip(*[oracle.warp_bwd(K, Kinv, pm[s], depth[s], src[s], gw[s], H, W) for s in range(S)])
    g_depth = np.stack(gd)[:, None] + 0.02 * greg / 80.0
    assert rel_to_max(g_depth[..., ::sub, ::sub], d["sub_g_depth"]) < 1e-4
    assert rel_to_max(np.stack(gp), d["g_pose_mats"]) < 1e-4
    assert rel_to_max(np.stack(gs)[..., ::sub, ::sub], d["sub_g_src"]) < 1e-4


def test_source_dims_differ_from_grid_dims(oracle):
    """normalisation uses the SOURCE dims; grid 24x80 sampled from a 48x160 source."""
    d = load_golden("srcdims_24x80_from_48x160")
    w = np.stack([oracle.warp(d["K"], d["Kinv"], d["pose_mats"][s], d["depth"][s], d["src"][s], 24, 80) for s in range(2)])
    assert np.array_equal(w, d["warped"])
    out, norm = oracle.grid_sample(d["src"][0], d["coords_px"][0])
    assert np.array_equal(out, d["warped_rank3"][0]) and np.array_equal(norm, d["coords_norm_rank3"])


def test_plane_sweep_cost_volume_softargmin(oracle):
    d = load_golden("plane_sweep_24x80")
    K, Kinv, pm, src = d["K"], d["Kinv"], d["pose_mats"], d["src"]
    synth = [oracle.synthesize_from_depths(K, Kinv, pm[s], d["depths"], 80, 24, src[s]) for s in range(2)]
    assert crc32(synth[0]) == int(d["crc_synth0"]) and crc32(synth[1]) == int(d["crc_synth1"])   # synthesis.py:57-80
    assert np.array_equal(synth[0][d["synth0_plane_ids"]], d["synth0_planes"])
    cv = np.stack([oracle.cost_volume(d["target_small"][0], s) for s in synth])
    assert np.array_equal(cv, d["cost_volumes"])                                                  # metrics.py:5-18
    assert np.abs(oracle.depth_from_costs(cv, True) - d["cand_sid"]).max() < 1e-5                 # synthesis.py:119-138
    assert np.abs(oracle.depth_from_costs(cv, False) - d["cand_uid"]).max() < 1e-5


RESIZE_CASES = (("a", "a_19x20", (19, 20)), ("a", "a_50x131", (50, 131)), ("a", "a_1x1", (1, 1)),
                ("b", "b_37x53", (37, 53)), ("b", "b_7x64", (7, 64)))


def test_resizes(oracle):
    """functional.interpolate as train.py:55 (bilinear) and :60 (bicubic) call it.  The vectors are torch CPU's; its
    vectorised sums associate differently from the scalar restatement: 2.5e-7 bilinear (one ulp), 2e-6 bicubic (16
    signed terms); north_star asks for 1e-5."""
    d = load_golden("resize_vectors")
    ps = load_golden("plane_sweep_24x80")
    assert np.abs(oracle.resize_bilinear(ps["target"], (24, 80)) - ps["target_small"][0]).max() < 2.5e-7
    assert np.abs(oracle.resize_bicubic(ps["cand_sid"], (48, 160)) - d["cand_full"]).max() < 2e-6
    for key, tag, size in RESIZE_CASES:
        assert np.abs(oracle.resize_bilinear(d[key], size) - d["bilinear_" + tag]).max() < 2.5e-7, tag
        assert np.abs(oracle.resize_bicubic(d[key], size) - d["bicubic_" + tag]).max() < 2e-6, tag


def test_depth_metrics(oracle):
    d = load_golden("metrics_3x40x96")
    m = oracle.depth_metrics(d["pred"], d["gt"])
    for k in ("rmse", "rmsle", "abs_rel", "sq_rel"):
        assert abs(m[k] - float(d[k])) < 1e-4 * abs(float(d[k])), k                               # metrics.py:100-153
    z = oracle.depth_metrics(d["pred"][:1], np.zeros_like(d["gt"][:1]))
    for k in ("rmse", "rmsle", "abs_rel", "sq_rel"):
        assert z[k] == float(d[k + "_allzero"]) == 0.0
    assert m["n_valid"] == float((d["gt"] > 0).sum())
    # delta<1.25^k is not in the reference (parity unpinned): check against a direct numpy statement
    p, g = d["pred"].astype(np.float64), d["gt"].astype(np.float64)
    valid = g > 0
    r = np.where(valid, np.maximum(p / np.where(valid, g, 1), np.where(valid, g, 1) / p), np.inf)
    for k, thr in (("delta1", 1.25), ("delta2", 1.25 ** 2), ("delta3", 1.25 ** 3)):
        ref = np.mean([(r[b] < thr).sum() / r[b].size for b in range(r.shape[0])])
        assert abs(m[k] - ref) < 2e-3
    # the counts behind them are integers: exact against a float32 numpy count (same IEEE divides, same thresholds)
    B = d["pred"].shape[0]
    p32, g32 = d["pred"].reshape(B, -1), d["gt"].reshape(B, -1)
    v32 = g32 > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        r32 = np.maximum(p32 / g32, g32 / p32)
    want = np.stack([(v32 & (r32 < np.float32(t))).sum(1) for t in (1.25, 1.5625, 1.953125)] + [v32.sum(1)], axis=1)
    got = oracle.depth_metric_counts(d["pred"], d["gt"])
    assert got.dtype == np.int64 and np.array_equal(got, want)
    for i, k in enumerate(("delta1", "delta2", "delta3")):
        assert abs(m[k] - (want[:, i] / p32.shape[1]).mean()) < 1e-12


RESIZE_U8_SIZES = [(37, 53, 19, 20), (19, 20, 37, 53), (64, 64, 64, 64), (90, 7, 3, 50), (5, 5, 11, 2), (192, 640, 96, 320),
                   (1, 1, 4, 4), (400, 300, 13, 17), (2, 3, 2, 3), (33, 1000, 33, 40)]


def test_pil_resize_restatement(oracle):
    """f.4 (dataset.py:51,110): the oracle's restatement of Pillow's antialiased BILINEAR resize, bit-exact (integer
    arithmetic) against (a) the resized frames the reference-generated fixtures hold, from the raw 1241x376 frame they were
    made of, and (b) PIL itself (a third-party package of this image, not the reference) on ragged sizes, up- and
    down-scaling."""
    raw = np.load(os.path.join(os.path.dirname(__file__), "golden", "raw_frame_376x1241_u8.npz"))["frame"]
    assert raw.shape == (376, 1241, 3) and raw.dtype == np.uint8
    for name, size in (("real_192x640", (192, 640)), ("real_48x160", (48, 160))):
        want = np.load(os.path.join(os.path.dirname(__file__), "golden", name + "_frames_u8.npz"))["frames"][0]
        assert np.array_equal(oracle.pil_resize_bilinear_u8(raw, size), want), name
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(0)
    for H, W, ho, wo in RESIZE_U8_SIZES:
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(a).resize((wo, ho), Image.BILINEAR))
        assert np.array_equal(oracle.pil_resize_bilinear_u8(a, (ho, wo)), want), (H, W, ho, wo)
    with pytest.raises(ValueError):
        oracle.pil_resize_bilinear_u8(np.zeros((100, 7, 3), np.uint8), (3, 50))   # 33x down: beyond the 64-tap limit


def test_plane_sweep_at_the_reference_shape(oracle):
    """train.py:49-60 as the reference runs it (constants.COST_* grid 96x320, sources 192x640, 80 SID planes): the oracle's
    plane images and cost volume bit for bit (CRC), the candidate depth to 1e-6 and its bicubic up-resize to 1e-5, against
    tests/golden/plane_sweep_96x320.npz (tools/gen_golden.py plane_sweep_full)."""
    d = load_golden("plane_sweep_96x320")
    fr = real_frames("real_192x640")
    src, tgt = fr[:-1], fr[-1]
    H, W = 96, 320
    pm = oracle.pose_from_vec(d["pose_vec"])
    assert np.abs(pm - d["pose_mats"]).max() <= 2e-7
    small = oracle.resize_bilinear(tgt, (H, W))
    cvs = []
    for s in range(3):
        syn = oracle.synthesize_from_depths(d["K"], d["Kinv"], d["pose_mats"][s], d["depths"], W, H, src[s])
        assert crc32(syn) == int(d["crc_synth"][s])                      # synthesis.py:57-80
        cvs.append(oracle.cost_volume(small, syn))
    cvs = np.stack(cvs)
    # the target at cost resolution comes from the oracle's interpolate (1e-6 of torch's): costs to 1e-6, not CRC
    assert np.abs(cvs[:, ::8, ::4, ::4] - d["sub_cost"]).max() < 1e-6    # metrics.py:5-18
    cand = oracle.depth_from_costs(cvs)
    assert np.abs(cand - d["cand"]).max() < 1e-5                         # synthesis.py:119-138
    full = oracle.resize_bicubic(d["cand"], (192, 640))
    assert np.abs(full[..., ::4, ::4] - d["sub_full"]).max() < 1e-5      # train.py:60


def test_facade_vectors(oracle):
    """rmsle with a non-default eps (metrics.py:114) and reprojection_loss on one prediction / distinct targets
    (metrics.py:54-72 with broadcasting) against the reference-made vectors."""
    g = load_golden("facade_vectors")
    m = load_golden("metrics_3x40x96")
    for tag, eps in (("1e-3", 1e-3), ("0.5", 0.5), ("0", 0.0)):
        want = float(g["rmsle_eps_" + tag])
        assert abs(oracle.depth_metrics(m["pred"], m["gt"], eps=eps)["rmsle"] - want) < 1e-5 * abs(want), tag
    d = load_golden("synth_oov_24x80")
    S, _, H, W = d["warped"].shape
    one = oracle.reprojection_loss(d["warped"][:1], d["target"])[0]
    assert np.array_equal(one, g["rank3_both"].reshape(H, W))
    per = [oracle.reprojection_loss(d["warped"][s:s + 1], g["tstack"][s])[0] for s in range(S)]
    assert np.array_equal(np.minimum.reduce(per), g["rank4_distinct_targets"].reshape(H, W))
    per0 = [oracle.reprojection_loss(d["warped"][:1], g["tstack"][s])[0] for s in range(S)]
    assert np.array_equal(np.minimum.reduce(per0), g["rank3_pred"].reshape(H, W))
    assert np.array_equal(oracle.ssim(d["warped"][:1], d["target"][None]), g["err_rank3"])
