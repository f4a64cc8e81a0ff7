"""plane-sweep kernel time for a given library build: sweep_lib_bench.py <lib.so>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import depth_estimation_amd._lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

import bench
import depth_estimation_amd as de

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["b12_s2_sweep80_96x320"]
sw = bench.build_problem(de, wl, dev, 0, False)[0]
st = torch.cuda.current_stream().cuda_stream
for _ in range(200):
    sw.calls["plane_sweep"](st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(300):
    sw.calls["plane_sweep"](st)
b.record()
torch.cuda.synchronize()
print(sys.argv[1], "plane_sweep %.1f us" % (a.elapsed_time(b) / 300 * 1e3))
