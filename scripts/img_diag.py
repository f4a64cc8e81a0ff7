import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import depth_estimation_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch, numpy as np
import bench
import depth_estimation_amd as de
dev = torch.device("cuda", 0)
sc = bench.Scale(de, 12, 2, 192, 640, dev, 0, True)
if len(sys.argv) > 2 and sys.argv[2] == "smooth":
    sc.depth.copy_(torch.linspace(80, 5, 192, device=dev).view(1, 1, 192, 1).expand_as(sc.depth) + 0.5 * torch.rand_like(sc.depth))
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    sc.step(s)
torch.cuda.synchronize()
g = sc.g_src.cpu().numpy().reshape(24, 3, 192, 640)
RR = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cyc = g[:, 0, ::RR, ::64]      # (24 images, 24 row groups, 10 chunks)
tiles = g[:, 1, ::RR, ::64]; pts = g[:, 2, ::RR, ::64]; hits = g[:, 0, 1::RR, ::64]
print("cycles: mean %.0f median %.0f p99 %.0f max %.0f" % (cyc.mean(), np.median(cyc), np.percentile(cyc, 99), cyc.max()))
print("by chunk: cycles", np.round(cyc.mean(axis=(0, 1))))
print("by chunk: point tiles", np.round(tiles.mean(axis=(0, 1)), 1))
print("by chunk: points scanned", np.round(pts.mean(axis=(0, 1))))
print("by chunk: hits", np.round(hits.mean(axis=(0, 1))))
order = np.argsort(cyc.ravel())[::-1][:8]
for o in order:
    i = np.unravel_index(o, cyc.shape)
    print("slow wave", i, "cycles %.0f tiles %.0f pts %.0f hits %.0f" % (cyc[i], tiles[i], pts[i], hits[i]))
print("corr cycles~pts %.2f  cycles~hits %.2f" % (np.corrcoef(cyc.ravel(), pts.ravel())[0, 1], np.corrcoef(cyc.ravel(), hits.ravel())[0, 1]))
