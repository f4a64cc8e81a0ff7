"""warp backward with the image gradient at B=12 192x640 for a given build of the library (A/B of tilings): python scripts/img_grad_rows.py <lib>"""
import os, sys, ctypes, time, torch
sys.path.insert(0, os.getcwd())
import depth_estimation_amd as de
import bench
lib_path = sys.argv[1]
de._lib._lib = None
de._lib.LIB_PATH = lib_path
dev = torch.device("cuda", 0)
sc = bench.Scale(de, 12, 2, 192, 640, dev, 0, True)
s = torch.cuda.Stream(); torch.cuda.set_stream(s); st = s.cuda_stream
sc.calls["warp_identity"](st); sc.calls["loss_fb"](st); torch.cuda.synchronize()   # a real d loss / d warped
for name in ("warp_bwd",):
    f = sc.calls[name]
    for _ in range(50): f(st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): f(st)
    torch.cuda.synchronize()
    print(lib_path, name, "%.1f us" % ((time.perf_counter() - t0) / 300 * 1e6))
