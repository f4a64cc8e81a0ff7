#!/usr/bin/env python3
"""profiles/pmc.json from the per-kernel PMC summaries of scripts/profile_round.sh:
    python scripts/pmc_to_json.py <tag>      reads gpurun_out/pmc_<tag>_{FETCH_SIZE,WRITE_SIZE,SQ1}_<workload>.txt
hbm_bytes = (FETCH_SIZE x 2 + WRITE_SIZE) KiB -> bytes per launch (x2: gfx950 counts the 128-B requests of coalesced
streaming reads at 64 B, MI355X_MICROARCH.md HBM; uncalibrated for gather kernels); valu_wave_insts = SQ_INSTS_VALU."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
# bench.py's call names -> the kernels a call launches (finalize kernels: a single wave, left out)
KERNELS = {"warp_identity": "warp_idt_kernel<2, true>", "loss_fb_warp_bwd": "photo_fb_kernel<2, true>",
           "loss_fb": "photo_fb_kernel<2, false>", "warp_bwd": "warp_bwd_kernel<3, false, 8>",
           "warp_fwd": "warp_fwd_kernel<3>", "loss_fwd": "photo_fwd_kernel<2, true>", "loss_bwd": "photo_bwd_kernel<2, true>",
           "metrics": "metrics_kernel", "plane_sweep": "plane_sweep_depth_kernel<3>", "resize_bilinear": "resize_bilinear_kernel",
           "resize_bicubic": "resize_bicubic_kernel"}


def read(path):
    out, cur = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+dispatches=(\d+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
        elif cur is not None and line.strip():
            k, v = line.split()
            cur[k] = float(v)
    return out


res = {"_comment": __doc__.strip().split("\n", 2)[2].strip() + f"  Source: gpurun_out/pmc_{tag}_* (copies under profiles/)."}
for wl in ("b12_s2_192x640_fwdbwd", "b64_s2_320x1024_fwdbwd", "b12_s2_sweep80_96x320"):
    f, w, q, q2 = [read(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{c}_{wl}.txt"))
                   for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ1", "SQ2")]
    d = {}
    for name, kern in KERNELS.items():
        e = {}
        if kern in f and kern in w:
            e["hbm_bytes"] = int((2 * f[kern]["FETCH_SIZE"] + w[kern]["WRITE_SIZE"]) * 1024)
        if kern in q and "SQ_INSTS_VALU" in q[kern]:
            e["valu_wave_insts"] = int(q[kern]["SQ_INSTS_VALU"])
        if kern in q2 and "SQ_INSTS_VALU_FMA_F32" in q2[kern]:   # the instruction mix, for the issue-cost model of bench.py
            e["valu_fma"] = int(q2[kern]["SQ_INSTS_VALU_FMA_F32"])
            e["valu_add"] = int(q2[kern]["SQ_INSTS_VALU_ADD_F32"])
            e["valu_trans"] = int(q2[kern]["SQ_INSTS_VALU_TRANS_F32"])
        if e:
            d[name] = e
    if d:
        res[wl] = d
# config 3: the finest level's launches, from the single-level passes with --image-grad
f, w, q = [read(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{c}_imggrad.txt")) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ1")]
d = {}
# (the warp backward with the image gradient is two launches: the backward itself with its staging, then the reducer)
for name, kerns in (("warp_identity", ("warp_idt_kernel<2, true>",)), ("loss_fb", ("photo_fb_kernel<2, false>",)),
                    ("warp_bwd", ("warp_bwd_kernel<3, true, 8>", "img_reduce_kernel"))):
    e = {}
    if all(k in f and k in w for k in kerns):
        e["hbm_bytes"] = int(sum(2 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"] for k in kerns) * 1024)
    if all(k in q and "SQ_INSTS_VALU" in q[k] for k in kerns):
        e["valu_wave_insts"] = int(sum(q[k]["SQ_INSTS_VALU"] for k in kerns))
    if e:
        d[name] = e
if d:
    res["b12_s2_192x640_pyramid4_fwdbwd"] = d
json.dump(res, open(os.path.join(ROOT, "profiles", "pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
