"""Differential fuzz of the fused step against the unfused kernels it replaces: random (B, S, H, W), depth ranges and poses;
ds_train_step_fwdbwd vs ds_warp_fwd + ds_fused_loss_fwd + ds_fused_loss_bwd + ds_warp_bwd + ds_smooth_batch_fwd/bwd.
Maps bit for bit, gradients to 1e-5 of the largest.   python scripts/fuzz_step.py [cases] [seed]"""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import depth_estimation_amd as de
L = de._lib
dev = "cuda:0"
p = lambda t: t.data_ptr()  # noqa: E731


def run(cases=200, seed=1, verbose=True):
    """-> number of cases with differences"""
    lib = L.lib()
    rng = random.Random(seed)
    bad = 0
    for it in range(cases):
        S = rng.choice([1, 2, 2, 2, 3, 4])
        H = rng.choice([2, 3, 5, 8, 17, 31, 32, 33, 48, 64, 96, rng.randint(2, 200)])
        W = rng.choice([2, 3, 61, 62, 63, 64, 65, 66, 124, 125, 126, 127, 128, 129, 160, 320, rng.randint(2, 700)])
        B = rng.choice([1, 1, 2, 3, 5, 12]) if H * W < 50000 else rng.choice([1, 2])
        g = torch.Generator().manual_seed(it)
        N = B * S
        src = torch.rand(B, S, 3, H, W, generator=g).to(dev)
        if rng.random() < 0.2:
            src = (src * 4).round() / 4          # many ties between sources and flat windows
        tgt = torch.rand(B, 3, H, W, generator=g).to(dev)
        if rng.random() < 0.2:
            tgt = src[:, 0].clone()              # static scene: identity loss of source 0 is zero
        dmax = rng.choice([2.0, 80.0, 1000.0])
        depth = (rng.choice([0.05, 1.0]) + dmax * torch.rand(B, S, H, W, generator=g)).to(dev)
        vec = (rng.choice([0.0, 0.01, 0.3]) * torch.randn(N, 6, generator=g)).to(dev)
        pose = de.synthesis.construct_pose_batch(vec, dev).reshape(B, S, 3, 4).contiguous()
        K, Kinv = [t.to(dev) for t in de.fused.intrinsics(W, H)]
        f = dict(dtype=torch.float32, device=dev)
        st = L.stream()
        mk = lambda: dict(warped=torch.empty(B, S, 3, H, W, **f), out=torch.empty(B, H, W, **f), idt=torch.empty(B, H, W, **f),
                          aux=torch.empty(B, H, W, dtype=torch.uint8, device=dev), sums=torch.empty(B, 2, **f),
                          gw=torch.empty(B, S, 3, H, W, **f), gd=torch.empty(B, S, H, W, **f), gp=torch.empty(N, 3, 4, **f))
        a, b = mk(), mk()
        reg, total = torch.empty(1, **f), torch.empty(1, **f)
        n = lib.ds_train_step_fwdbwd_workspace_bytes(B, S, H, W)
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        L.check(lib.ds_train_step_fwdbwd(p(K), p(Kinv), p(pose), p(depth), p(src), p(tgt), 0.02, B, S, H, W, p(a["warped"]), p(a["out"]),
                                         p(a["idt"]), p(a["aux"]), p(a["sums"]), p(reg), p(total), p(a["gw"]), p(a["gd"]), p(a["gp"]),
                                         p(ws), n, st), "train_step")
        n1 = max(lib.ds_fused_loss_workspace_bytes(B, H, W), lib.ds_warp_bwd_workspace_bytes(N, H, W), lib.ds_smooth_workspace_bytes(N, H, W))
        w1 = torch.empty(n1, dtype=torch.uint8, device=dev)
        L.check(lib.ds_warp_fwd(p(K), p(Kinv), p(pose), p(depth), p(src), N, 3, H, W, H, W, p(b["warped"]), st), "warp_fwd")
        L.check(lib.ds_fused_loss_fwd(p(b["warped"]), p(src), p(tgt), B, S, H, W, p(b["out"]), p(b["idt"]), p(b["aux"]), p(b["sums"]),
                                      p(w1), n1, st), "loss_fwd")
        L.check(lib.ds_fused_loss_bwd(p(b["warped"]), p(tgt), p(b["idt"]), p(b["aux"]), 0, B, S, H, W, p(b["gw"]), st), "loss_bwd")
        L.check(lib.ds_warp_bwd(p(K), p(Kinv), p(pose), p(depth), p(src), p(b["gw"]), N, 3, H, W, H, W, p(b["gd"]), p(b["gp"]), 0,
                                p(w1), n1, st), "warp_bwd")
        g_reg = torch.full((1,), 0.02, **f)
        reg2 = torch.empty(1, **f)
        L.check(lib.ds_smooth_batch_fwd(p(depth), p(tgt), B, S, 3, H, W, p(reg2), p(w1), n1, st), "smooth_fwd")
        L.check(lib.ds_smooth_batch_bwd(p(depth), p(tgt), p(g_reg), B, S, 3, H, W, 1, p(b["gd"]), st), "smooth_bwd")
        torch.cuda.synchronize()
        msgs = []
        for k in ("warped", "out", "idt", "aux"):
            if not torch.equal(a[k], b[k]):
                msgs.append(f"{k}: {int((a[k] != b[k]).sum())} differ")
        for k in ("sums", "gd", "gp"):
            den = float(b[k].abs().max()) or 1.0
            e = float((a[k] - b[k]).abs().max()) / den
            if not (e < 1e-5) or not bool(torch.isfinite(a[k]).all()):
                msgs.append(f"{k}: rel {e:.2e}")
        if abs(float(reg) - float(reg2)) > 1e-5 * abs(float(reg2)):
            msgs.append(f"reg {float(reg)} vs {float(reg2)}")
        if msgs:
            bad += 1
            print(f"case {it}: B={B} S={S} {H}x{W}: " + "; ".join(msgs), flush=True)
    return bad


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bad = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"{cases} cases, {bad} with differences")
    sys.exit(1 if bad else 0)
